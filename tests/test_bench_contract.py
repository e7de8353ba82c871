"""CPU: the committed bench line (profiles/r<NN>_bench_latest.json of the newest round, written by bench.py
on an MI355X) carries every field of the driver's contract, with consistent arithmetic; and the PMC
traffic it quotes comes from a pass taken on the kernel sources that are in the tree."""
import glob
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def _latest(name):
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + name)))[-1]


def test_recorded_bench_line_has_the_contract_fields():
    line = open(_latest("bench_latest.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic_source" in r                      # where the PMC figure comes from, or why there is none
    # achieved = algorithmic bytes per launch / measured launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    # PMC traffic is per launch and close to the algorithmic bytes for this streaming kernel
    assert r["traffic"] is None or 0.99 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # value = elements of all ranks / wall time, consistent with ms_per_step
    per_step = d["config"]["elements_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]


def test_batch_inversion_is_reported_against_the_valu_roofline():
    d = json.loads(open(_latest("bench_latest.json")).read().strip().splitlines()[-1])
    inv = d["extra"]["batch_inv_bls12_381"]
    assert inv["bound"] == "valu" and inv["mads_per_element"] == 900
    assert abs(inv["valu_frac"] - inv["valu_mad_Tops_per_gpu"] / 39.3) < 1e-6
    assert "valu_kernel_ms_for_comparison" not in d["extra"]["matmul_goldilocks"]      # timed live now
    assert d["extra"]["matmul_goldilocks"]["valu_kernel_ms_same_run"] > 50
    assert "bf16" in d["extra"]["matmul_goldilocks"]["int8_peak_note"]      # the int8 peak is priced from the guide's bf16 figure
    mm = d["extra"]["matmul_bn254"]                                         # multi-limb product on the matrix cores beside the VALU kernel
    assert mm["same_bits_as_valu_kernel"] is True and mm["ms_per_matmul"] < mm["valu_kernel_ms_same_run"]
    assert abs(mm["mfma_frac"] - 2 * 64 * 16 * mm["dim"] ** 3 / (mm["ms_per_matmul"] * 1e-3) / 1e12 / mm["int8_mfma_peak_Tops"]) < 1e-9


def test_bare_multi_gpu_invocation_spawns_its_ranks():
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE) must not die in argument handling: it
    starts the N ranks as a fresh child process under torch.distributed.run (one rank per GPU, loopback
    rendezvous) and relays the child's line.  --dry-run-launch prints the command instead of running it."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run-launch"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["would_launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]      # the child sees the caller's flags, not the dry-run one
    # one GPU never spawns: the N = 1 line stays the plain single-process run (here it stops at "no device")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run-launch"],
                        capture_output=True, text=True, env=env, timeout=300)
    assert "would_launch" not in r1.stdout


def test_pmc_traffic_is_quoted_only_from_a_pass_on_the_same_kernel_sources(tmp_path, monkeypatch):
    """bench.py quotes profiles/r*_pmc_traffic_*.json only when the pass measured the kernel sources that
    are in the tree now (each file carries their SHA-256): a fresh file gives the figure and names its
    commit, a stale one gives None and says so, a pass over another problem size gives None."""
    import bench
    sha = bench.kernel_sources_sha256()
    for w, key, val in (("mul", "elements", 1 << 28), ("matvec", "rows", 1 << 15), ("matmul", "dim", 8192),
                        ("batch_inv", "elements", 1 << 24)):
        pmc = json.load(open(_latest("pmc_traffic_%s.json" % w)))
        traffic, src = bench._pmc_traffic(w, key, val)
        if pmc["kernel_sources_sha256"] == sha:
            assert traffic and traffic >= 0.99 * pmc["algorithmic_bytes_per_launch"] and "same kernel sources" in src
        else:   # collected before the last kernel change: must not be quoted
            assert traffic is None and "stale" in src
        assert bench._pmc_traffic(w, key, 12345)[0] is None
    # the same logic on a synthetic pair of files, so that both branches are always exercised
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "crypto_primitives_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "crypto_primitives_amd" / "csrc" / "k.cuh").write_text("kernel v1")
    fresh = {"elements": 8, "traffic_bytes_per_launch": 777.0, "commit": "abc1234", "kernel_sources_sha256": bench.kernel_sources_sha256()}
    (tmp_path / "profiles" / "r9_pmc_traffic_mul.json").write_text(json.dumps(fresh))
    assert bench._pmc_traffic("mul", "elements", 8) == (777.0, "profiles/r9_pmc_traffic_mul.json (collected at commit abc1234, same kernel sources as this tree)")
    (tmp_path / "crypto_primitives_amd" / "csrc" / "k.cuh").write_text("kernel v2")
    t, src = bench._pmc_traffic("mul", "elements", 8)
    assert t is None and "stale" in src and "abc1234" in src
