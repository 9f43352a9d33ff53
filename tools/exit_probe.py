import subprocess, time, os, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
def wall(cmd, env=None):
    t=time.perf_counter(); p=subprocess.run(cmd,capture_output=True,text=True,env=env); return time.perf_counter()-t, p
for i in range(4):
    w,p=wall([ROOT+"/tools/ubench/hip_startup"])
    tot=[l for l in p.stdout.splitlines() if l.startswith("total since main")]
    print("hip_startup: process %.3f s; %s" % (w, tot[-1] if tot else p.stdout[-200:]))
