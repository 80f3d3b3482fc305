"""CPU suite: host logic (group lists, sharding, the world-size-2 gather over gloo)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from autoinst_amd import sharding, synth
from autoinst_amd.ncuts_api import _groups_from_labels


def test_groups_from_labels_order_and_members():
    lab = np.array([2, 0, 1, 0, 2, 2], dtype=np.int32)
    ids = np.arange(10, 16)
    g = _groups_from_labels(lab, 3, ids)
    assert [x.tolist() for x in g] == [[11, 13], [12], [10, 14, 15]]


def test_lpt_assign_balanced_and_complete():
    sizes = [200_000] * 8 + [int(x) for x in np.geomspace(3000, 30000, 64)]
    parts = sharding.lpt_assign(sizes, 8)
    assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
    load = [sum(sharding.chunk_cost(sizes[i]) for i in p) for p in parts]
    assert max(load) / min(load) < 1.05
    assert sharding.lpt_assign(sizes, 8) == parts


def test_synth_is_deterministic_and_voxel_unique():
    p1, g1 = synth.surface_chunk(3000, seed=3)
    p2, g2 = synth.surface_chunk(3000, seed=3)
    assert np.array_equal(p1, p2) and np.array_equal(g1, g2)
    key = np.floor(p1 / synth.VOXEL).astype(np.int64)
    assert np.unique(key, axis=0).shape[0] == 3000
    f = synth.surrogate_features(g1, 96, 3)
    assert f.shape == (3000, 96) and 0.02 < (~f.any(1)).mean() < 0.09
    assert np.array_equal(f, f.astype(np.float32).astype(np.float64))


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from autoinst_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
sizes = [50, 7, 31, 12, 5]
mine = sharding.lpt_assign(sizes, dist.get_world_size())[rank]
local = {i: (np.arange(sizes[i], dtype=np.int32) * (i + 1)) % 11 for i in mine}
out = sharding.gather_labels(local)
if rank == 0:
    assert sorted(out) == list(range(len(sizes))), sorted(out)
    for i in range(len(sizes)):
        assert np.array_equal(out[i], (np.arange(sizes[i], dtype=np.int32) * (i + 1)) % 11)
    print("GATHER_OK")
else:
    assert out is None
dist.destroy_process_group()
"""


def test_gather_labels_world_size_2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


_WORKER4 = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from autoinst_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
# configs[2]'s chunk-size mix (3k-30k points, log-uniform: an UNEVEN deal), three steps through the asynchronous gather as bench.py
# issues it (step s + 1's exchange before step s's result is waited for); rank 3 holds NOTHING in step 1 (an empty payload)
rng = np.random.default_rng(5)
sizes = np.exp(rng.uniform(np.log(3000), np.log(30000), 13)).astype(int).tolist()
deal = sharding.lpt_assign(sizes, world)
assert sorted(c for d in deal for c in d) == list(range(len(sizes)))
pending, outs = None, []
for step in range(3):
    mine = [] if (step == 1 and rank == 3) else deal[rank]
    local = {c: ((np.arange(sizes[c], dtype=np.int32) + step) % (c + 2)).astype(np.int32) for c in mine}
    fut = sharding.gather_labels_async(local)
    if pending is not None:
        outs.append(pending.result() if rank == 0 else None)
    pending = fut
    assert (fut is None) == (rank != 0)
outs.append(pending.result() if rank == 0 else None)
if rank == 0:
    for step, out in enumerate(outs):
        want = [c for r, d in enumerate(deal) for c in d if not (step == 1 and r == 3)]
        assert sorted(out) == sorted(want), (step, sorted(out))
        for c in want:
            assert np.array_equal(out[c], (np.arange(sizes[c], dtype=np.int32) + step) % (c + 2)), (step, c)
    print("GATHER4_OK", [len(d) for d in deal])
dist.barrier()
dist.destroy_process_group()
"""


def test_gather_labels_async_world_size_4_uneven_mix_and_an_empty_rank(tmp_path):
    """World size 4 over gloo: the uneven cfg3 chunk mix dealt by LPT, three pipelined steps through `gather_labels_async`, one
    rank's payload empty in one step (a rank that ran out of chunks must still take part in both collectives)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w4.py"
    script.write_text(_WORKER4)
    procs = []
    for r in range(4):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER4_OK" in outs[0]


def test_run_chunks_argument_checks():
    """`sharding.run_chunks` without work does not touch the GPU; bad thread / batch counts raise."""
    assert sharding.run_chunks([]) == []
    with pytest.raises(ValueError):
        sharding.run_chunks([(np.zeros((3, 3)), None)], threads=0)
    with pytest.raises(ValueError):
        sharding.run_chunks([(np.zeros((3, 3)), None)], batch=0)
