import subprocess, time, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
for mode,mib,n in ((0,0,0),(1,40,3),(2,40,3),(3,40,3),(1,120,1),(2,120,1),(1,40,6),(2,40,6)):
    rows=[]
    for k in range(4):
        time.sleep(0.4)
        t=time.perf_counter(); p=subprocess.run([R+"/tools/ubench/pin_probe",str(mode),str(mib),str(n)],capture_output=True,text=True); w=time.perf_counter()-t
        rows.append((w,p.stdout.strip()))
    rows.sort()
    print("process %.3f s (min of 4: %.3f)  %s" % (rows[len(rows)//2][0], rows[0][0], rows[len(rows)//2][1]), flush=True)
