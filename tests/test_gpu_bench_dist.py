"""GPU: bench.py's multi-rank code path on the one GPU of the box.  Launched the way the driver
launches N > 1 (torch.distributed.run, one rank per GPU) with one rank, bench.py initialises the
RCCL process group, all-gathers the mat-vec output and the mat-mul product with
`all_gather_into_tensor`, and takes the max over ranks with an all-reduce -- the branches the N = 1
default run never enters.  Reduced sizes: this is a rehearsal of the plumbing, not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload", ["matvec", "matmul", "batch_inv", "mul"])
def test_bench_distributed_branch_runs_on_one_rank(workload):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--log2n", "20", "--matvec-log2", "11", "--matmul-dim", "1024", "--inv-log2", "18",
           "--workload", workload]
    if workload != "mul":
        cmd.append("--no-extra")       # the mul run keeps the extras: every secondary workload under the process group
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["distributed"]["process_group"] == "nccl (RCCL)" and line["distributed"]["world"] == 1
    if workload == "matvec":
        assert "allgather" in line["config"]["parallelism"]
    if workload == "matmul":
        assert line["config"]["ms_per_step_with_allgather_of_C"] > 0      # the all-gather of C ran
    if workload == "mul":
        assert line["extra"]["matmul_goldilocks"]["ms_per_matmul_with_allgather_of_C"] > 0
        assert line["extra"]["batch_inv_bls12_381"]["x_times_inv_is_one"] is True


def test_in_library_sharded_block_with_eight_entries_on_one_gpu():
    """extra.in_library_sharded: SURVEY.md 8e's own design (one process, fpgpu_set_devices, sharded / replicated
    handles, all-gather of the mat-vec's row blocks) timed by bench.py, rehearsed with a device list that names
    the box's GPU eight times (eight streams, eight ranges / row blocks, peer-copy all-gather)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu",
           "--log2n", "20", "--matvec-log2", "11", "--matmul-dim", "1024", "--inv-log2", "18",
           "--sharded-devices", "0,0,0,0,0,0,0,0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    blk = line["extra"]["in_library_sharded"]
    assert "error" not in blk, blk
    assert blk["devices"] == [0] * 8 and "peer copies" in blk["transport"]
    assert blk["mul_bn254"]["Gmul_per_s"] > 0 and blk["batch_inv_bls12_381"]["x_times_inv_is_one_on_sample"] is True
    mv = blk["matvec_bn254"]
    assert mv["gathered_equals_sharded"] is True and mv["ms_y_replicated_with_allgather"] > 0 and mv["ms_y_row_sharded"] > 0
