"""The bench line committed under profiles/ carries every field the contract names."""
import json
import os

from conftest import ROOT


def _newest_bench_line():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_200k_line.json")))[-1]


def test_committed_bench_line_has_the_contract_fields():
    """The newest `profiles/r*_bench_200k_line.json` (a `python bench.py` run on an MI355X box)."""
    with open(_newest_bench_line()) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "chunks/sec" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["chunks_per_step"] * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    if "r01" not in _newest_bench_line() and "r02" not in _newest_bench_line():
        # since round 3: the pool is min(host cores, 128) workers and says which host the cached full-size run came from
        assert c["pool"]["cores"] == min(c["pool"]["host_cores_available"], 128) and "cpu_model" in c["cached_host"]
        assert r["kernel"] == "fk_spmv" and (r["traffic"] is None or r["traffic"] > 0)


def test_bench_launches_its_own_ranks_dry():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts the two ranks itself (gloo rehearsal, no GPU)
    and prints exactly one JSON line; a WORLD_SIZE that disagrees with --gpus is refused."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]   # gloo itself prints a connection notice
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["dry"] is True and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["scaling"] == "weak"
    # every rank's start-up record is in rank 0's line, and the first collective was bounded and timed
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and all(r["cpus"] >= 1 and r["pid"] > 0 for r in line["ranks"])
    assert line["first_collective_ms"] is not None and line["first_collective_ms"] >= 0.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_bench_self_launch_propagates_a_failing_rank():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    # without --dry the ranks need a GPU: each refuses to start, and the parent must report that
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "exited with code" in p.stderr and "rank" in p.stderr   # the parent names the rank that failed


def test_first_collective_that_never_completes_ends_the_rank_non_zero_and_names_it(tmp_path):
    """One of two ranks never arrives: the rank that is there must not hang in its first collective -- the watchdog ends it with a
    message that names the rank (here the rendezvous itself times out first or the watchdog fires: either way non-zero, bounded)."""
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = (
        "import os, sys, time, threading\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist, datetime\n"
        "import bench\n"
        "dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2, timeout=datetime.timedelta(seconds=120))\n"
        "if dist.get_rank() == 1:\n"
        "    time.sleep(60)\n"      # arrives at the rendezvous but never at the collective
        "    os._exit(0)\n"
        "bench.first_collective(dist, torch, torch.device('cpu'), 0, 2, 3.0, 'gloo')\n"
        "print('NOT REACHED')\n"
    )
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    out0, err0 = procs[0].communicate(timeout=120)
    took = time.time() - t0
    procs[1].kill()
    procs[1].wait()
    assert procs[0].returncode == 75, (procs[0].returncode, err0[-1500:])
    assert "rank 0 of 2" in err0 and "did not complete" in err0 and "NOT REACHED" not in out0
    assert took < 60
