"""The one-process-per-GPU harness on CPU: gloo, world_size 2 (127.0.0.1 rendezvous)."""
import os
import socket
import subprocess
import tempfile
import sys

from phnrec_amd import distrun
from tests.util import ROOT

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from phnrec_amd import distrun
r = distrun.Ranks(gpus=2).init("gloo")
assert r.world == 2 and r.pg
lengths = [300, 1500, 700, 20, 999, 1, 450, 1200, 640]
owner = distrun.shard_by_frames(lengths, r.world)
mine = [i for i, o in enumerate(owner) if o == r.rank]
state = {"n": 0}
def step():
    time.sleep(0.01 * (1 + r.rank))       # rank 1 is slower: MAX over ranks must see it
    state["n"] += sum(lengths[i] for i in mine)
t = distrun.timed_steps(r, step, lambda: None, steps=5, warmup=1)
total = r.sum_int(state["n"])
print(json.dumps({"rank": r.rank, "t": t, "total": total, "mine": mine}))
r.finish()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_timing_and_sharding(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    assert outs[0]["t"] == outs[1]["t"], "every rank must report the MAX over ranks"
    assert outs[0]["t"] >= 5 * 0.02 * 0.9, "the slower rank (20 ms/step) sets the time"
    lengths = [300, 1500, 700, 20, 999, 1, 450, 1200, 640]
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(9)), "every utterance owned exactly once"
    assert outs[0]["total"] == outs[1]["total"] == 6 * sum(lengths)   # 1 warmup + 5 timed steps


def test_shard_by_frames_balance_and_determinism():
    import random
    rnd = random.Random(3)
    lengths = [rnd.randint(300, 1500) for _ in range(10000)]     # BASELINE configs[3] list
    for world in (1, 2, 4, 8):
        owner = distrun.shard_by_frames(lengths, world)
        assert owner == distrun.shard_by_frames(lengths, world)
        loads = [sum(l for l, o in zip(lengths, owner) if o == r) for r in range(world)]
        assert max(loads) - min(loads) <= 1500, loads
        assert sum(loads) == sum(lengths)
    assert distrun.shard_by_frames([], 4) == []
    assert set(distrun.shard_by_frames([5], 8)) == {0}


def test_single_process_is_a_noop_world():
    r = distrun.Ranks(gpus=1)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    assert r.world == env_world
    if env_world == 1:
        r.init("gloo")
        assert not r.pg
        assert r.max_float(1.5) == 1.5 and r.sum_int(7) == 7
        r.barrier()
        t = distrun.timed_steps(r, lambda: None, lambda: None, steps=3, warmup=1)
        assert t >= 0


# ---- bench.py's own launcher (python bench.py --gpus N without torch.distributed.run) ----------------------
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(text):
    import json
    return json.loads([l for l in text.strip().splitlines() if l.startswith("{")][-1])


def test_bench_self_launch_starts_n_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts two fresh ranks itself; the line
    rank 0 prints carries the world size the process group counted (stub step, gloo: no GPU here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub",
                        "--detail-out", os.path.join(tempfile.mkdtemp(), "detail.json")],
                       capture_output=True, text=True, env=env, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    # stdout is the record and nothing else (gloo announces its ranks on stdout from C++: bench.py points fd 1 at stderr)
    assert [l for l in p.stdout.splitlines() if l.strip()] == [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1, p.stdout[:500]
    assert len(p.stdout.strip()) + 1 <= 6144, "the stdout line stays inside what the driver parses"
    assert "bench detail [ranks]" in p.stderr, "the full record goes to stderr, leg by leg"
    line = _last_json(p.stdout)
    assert line["ranks"] == {"world": 2, "launcher": "self", "backend": "gloo", "device_map": [0, 1]}
    assert line["frames_all_ranks"] == 2 * (4 + 1) * 8192          # both ranks stepped, warm-up included
    assert line["ms_per_step"] >= 4.0 * 0.9, "rank 1 (4 ms per step) sets the time: MAX over ranks"


def test_bench_under_torch_distributed_run():
    """the driver's form: torch.distributed.run starts the ranks, bench.py does not start more"""
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1, p.stdout[:500]      # the record, alone
    line = _last_json(p.stdout)
    assert line["ranks"]["world"] == 2 and line["ranks"]["launcher"] == "torch.distributed.run"
    assert line["frames_all_ranks"] == 2 * (3 + 1) * 8192


def test_bench_eight_ranks_in_the_drivers_exact_form():
    """Pre-flight of the driver's scaling run at N = 8, as far as a box without eight GPUs allows: the driver's exact
    command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 8 --steps 20 --warmup 5` -- with --stub in place of the GPU step (gloo instead of RCCL; a one-GPU box
    admits six processes on its card, so eight GPU ranks cannot be rehearsed there either).  Everything else is the real
    path: rendezvous, the barriers around the timed region (the contract's bracket: the clock stops behind the closing barrier), MAX over
    ranks, SUM of the frames, one line from rank 0 -- well inside the driver's 600 s."""
    import time
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    t0 = time.time()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5", "--stub"],
                       capture_output=True, text=True, env=env, timeout=580)
    dt = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1, p.stdout[:500]
    line = _last_json(p.stdout)
    assert line["ranks"]["world"] == 8 and line["ranks"]["launcher"] == "torch.distributed.run"
    assert line["ranks"]["device_map"] == list(range(8)) and line["steps"] == 20 and line["warmup"] == 5
    assert line["frames_all_ranks"] == 8 * (20 + 5) * 8192
    # rank 7 sleeps 16 ms per step: MAX over ranks, the contract's bracket (closing barrier inside the clock)
    assert 16.0 * 0.9 <= line["ms_per_step"] < 16.0 * 1.5
    assert dt < 300, "took %.0f s" % dt


def test_bench_refuses_to_misreport_the_gpu_count():
    """More GPUs asked for than the node has (none here): a loud failure, never an n_gpus: 1 line; and a
    launcher's WORLD_SIZE that disagrees with --gpus is refused as well."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PHNREC_DEVICE_MAP")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert p.returncode != 0 and "refusing" in p.stderr and "{" not in p.stdout
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--stub"], capture_output=True, text=True, env=env2, timeout=240)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_device_map_override(monkeypatch):
    monkeypatch.delenv("PHNREC_DEVICE_MAP", raising=False)
    assert distrun.device_map(4) == [0, 1, 2, 3]
    monkeypatch.setenv("PHNREC_DEVICE_MAP", "0,0")
    assert distrun.device_map(2) == [0, 0]
    import pytest
    with pytest.raises(SystemExit):
        distrun.device_map(3)


# ---- the stdout line of bench.py: bounded, numbers only (phnrec_amd/benchline.py) ------------------------
def test_bench_line_of_the_round5_record_fits_six_kilobytes():
    """Round 5's full record (21.7 KB on stdout: the driver's parser cut it, BENCH_r05.parsed = null) through the
    compaction bench.py now applies: one line <= 6 KB that still carries the contract's keys, `roofline` with its cold
    figure, windows and traffic, `cpu_baseline` with the reference's three regimes, the port, the host and both parities,
    and every side leg as numbers; no prose; nothing dropped."""
    import json
    from phnrec_amd import benchline
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_form.json")))
    assert len(json.dumps(full)) > 20000
    line = benchline.stdout_line(full)
    assert len(line) + 1 <= benchline.LIMIT == 6144
    c = json.loads(line)
    for k in benchline.CONTRACT + ("roofline", "cpu_baseline"):
        assert k in c, k
    for k in benchline.CONTRACT:
        assert c[k] == full[k] or k == "config", k          # the contract's keys travel unrounded
    assert c["config"]["workload"] == full["config"]["workload"]
    assert "dropped" not in c
    r = c["roofline"]
    assert r["frac"] == 0.8289 and r["kernel_ms"] == 0.1923 and r["cold"]["frac"] == 0.5083 and r["traffic"] == 64106436
    assert r["windows"]["kernel_ms_median"] == 0.1916 and r["traffic_source"] == "profiles/hbm_traffic.json"
    b = c["cpu_baseline"]
    assert b["kind"] == "reference" and b["cores"] == 1 and b["sgemm_all_cores"]["cores"] == 16 and b["port"]["kind"] == "port"
    assert b["host"]["cpu_model"].startswith("AMD EPYC") and b["parity_max_abs_vs_gpu"] < 1e-4
    sl = c["sharded_list"]
    assert set(sl) >= {"host", "E", "E_D", "F", "F_D", "weak_list", "host_ceiling", "mlf_all_modes_equal"}
    assert sl["weak_list"]["g8_default"]["mode"] == "F+D,auto" and sl["weak_list"]["F_D"]["ceiling_over_8_gpus"] > 1

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                assert k not in ("what", "cpu_s_by_stage", "create_trace_ms"), k
                walk(v)
        elif isinstance(o, str):
            assert len(o) <= 160, o
    walk(c)


def test_bench_line_gives_up_side_legs_before_the_contract():
    """a record whose compact form is still over the limit loses whole side legs, least important first, and says
    which; the contract's keys, `roofline` and `cpu_baseline` stay"""
    import json
    from phnrec_amd import benchline
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_form.json")))
    line = benchline.stdout_line(full, limit=3500)
    c = json.loads(line)
    assert len(line) + 1 <= 3500 and c["dropped"][0] == "push_bunch512" and "sharded_list" in c
    assert "roofline" in c and "cpu_baseline" in c and all(k in c for k in benchline.CONTRACT)
    import pytest
    with pytest.raises(RuntimeError):
        benchline.stdout_line(full, limit=500)
