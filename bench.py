#!/usr/bin/env python3
"""bench.py -- MI355X benchmark of the prime-field Montgomery hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
on rank 0.  For N > 1 it is launched under torch.distributed.run, one rank per GPU.

Headline workload (BASELINE.json configs[1]): ConstMontyField 256-bit (BN254 Fr)
batched Montgomery multiplication over 2^28 HBM-resident elements per GPU.  A "step"
is one pass `out[i] = mul(a[i], b[i])` over the whole batch; inputs are synthetic
uniform field elements generated on the device and are resident
in HBM before the timed region.  Elementwise batches shard across GPUs with no
data-path collective (weak scaling: every rank owns its own 2^28 elements).

The same line carries:
  roofline     -- the mul kernel against the HBM roofline (96 algorithmic bytes per
                  element: two 32-byte operands read, one 32-byte result written),
                  launch duration measured live with HIP events on the launch stream.
  cpu_baseline -- the CPU restatement of the reference path (oracle/, kind "port")
                  timed on this box's host cores on a bounded sample (2^20-element
                  batches = BASELINE.json configs[0] shape).
  extra        -- sqr / add / sub over the same batch, the register-resident dependent
                  mul chain (VALU-bound Gmul/s), BLS12-381 batch inversion and the
                  256-bit mat-vec (row-sharded, RCCL all-gather of y when N > 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0      # MI355X_MICROARCH.md matrix-core table: I8 16x16x64 / 32x32x32 = 2x BF16 per clock; dense bf16 ~2.5 PFLOP/s
MFMA_I8_PEAK_NOTE = ("int8 peak = 2x the guide's ~2.5 PFLOP/s dense bf16 figure (its matrix-core table: I8 runs at the cycles "
                     "of the BF16 form at 2x the K, i.e. 2x BF16 per clock); measured bare v_mfma_i32_16x16x64_i8 loop: "
                     "8.4 ns per MFMA per SIMD = 3.99 POP/s")
VALU_MAD_PEAK_TOPS = 39.3       # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz v_mad_u64_u32 lane-ops/s (measured 36.5)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=28, help="elements per GPU = 2^log2n")
    ap.add_argument("--workload", default="mul", choices=["mul", "sqr", "add", "sub", "matvec", "batch_inv", "matmul"])
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--matvec-log2", type=int, default=15, help="mat-vec is 2^k x 2^k")
    ap.add_argument("--matmul-dim", type=int, default=8192, help="mat-mul is dim^3 (rehearsals only; the config is 8192)")
    ap.add_argument("--inv-log2", type=int, default=24, help="batch inversion over 2^k elements in total")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (RCCL process group, all-gather, max over ranks) even at world size 1")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the torch.distributed.run command instead of running it")
    ap.add_argument("--sharded-devices", default=None,
                    help="device list of the in-library sharded measurement (extra.in_library_sharded), e.g. 0,0,0,0,0,0,0,0 to "
                         "rehearse eight parts on one GPU; default: 0..N-1")
    ap.add_argument("--no-sharded", action="store_true", help="skip extra.in_library_sharded")
    ap.add_argument("--sharded-only", action="store_true",
                    help="measure extra.in_library_sharded alone and print it as one JSON object (how the main run obtains it)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` invoked bare (no WORLD_SIZE) for N > 1: start the N ranks as a FRESH child process
    under torch.distributed.run (this process has not touched HIP and never replaces itself), relay the child's
    one JSON line and its return code."""
    import socket
    import subprocess
    with socket.socket() as sock:   # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--dry-run-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    if args.dry_run_launch:
        print(json.dumps({"would_launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on these hosts
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the launched ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    args = parse()
    if args.sharded_only:
        devs = [int(t) for t in args.sharded_devices.split(",")] if args.sharded_devices else list(range(args.gpus))
        print(json.dumps({"in_library_sharded": bench_in_library_sharded(devs)}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import crypto_primitives_amd as cpa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:   # the launcher's world size is what runs; say so, do not die on it
        print("bench.py: --gpus %d but WORLD_SIZE=%d: measuring %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the RCCL path is taken even for one rank, so that a one-GPU box can
    # rehearse exactly the code the N > 1 runs execute (process group, all-gather, max over ranks)
    distributed = world > 1 or args.force_dist or (world == 1 and "TORCHELASTIC_RUN_ID" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n = 1 << args.log2n
    f = cpa.MontyField(cpa.BN254_FR, 4)
    a = f.random_dev(n, 0x9E3779B97F4A7C15 ^ (2 + 1000 * rank))
    b = f.random_dev(n, 0x9E3779B97F4A7C15 ^ (3 + 1000 * rank))
    out = f.alloc(n)

    ops = {
        "mul": (lambda: f.mul_dev(a, b, out), 96),
        "sqr": (lambda: f.sqr_dev(a, out), 64),
        "add": (lambda: f.add_dev(a, b, out), 96),
        "sub": (lambda: f.sub_dev(a, b, out), 96),
    }

    def time_steps(fn, steps, warmup):
        """Returns (wall seconds for `steps` steps, max over ranks; mean launch duration in s from
        HIP events recorded on the launch stream)."""
        for _ in range(warmup):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            fn()
            e1.record()
        barrier()
        wall = time.perf_counter() - t0
        kern = sum(e0.elapsed_time(e1) for e0, e1 in evs) / steps * 1e-3
        return max_over_ranks(wall), max_over_ranks(kern)

    elementwise = args.workload in ops     # the headline family: every rank owns its own batch (weak scaling)
    result = {}
    extra = {}
    distributed_done = False
    samples = {}   # small host copies of GPU results; checked against the oracle in the CPU leg only

    # ---------------- primary workload ------------------------------------------------
    if args.workload in ops:
        fn, bytes_per = ops[args.workload]
        wall, kern = time_steps(fn, args.steps, args.warmup)
        total_elems = n * world * args.steps
        value = total_elems / wall / 1e9
        # a strided sample of the last pass, kept for the CPU leg below to check against the oracle
        # (the oracle is only touched there)
        verified = None
        sample = None
        if rank == 0:
            idx = torch.arange(0, n, max(1, n // 4096), device="cuda")
            sample = (args.workload, a[idx].cpu().numpy().view(np.uint64), b[idx].cpu().numpy().view(np.uint64),
                      out[idx].cpu().numpy().view(np.uint64))
        unit = {"mul": "Gmul/s", "sqr": "Gsqr/s", "add": "Gadd/s", "sub": "Gsub/s"}[args.workload]
        achieved = bytes_per * n / kern / 1e9
        # HBM bytes per launch from the PMC passes kept under profiles/ (collected with
        # rocprofv3 --pmc on this same command; bench.py cannot run the profiler on itself)
        traffic, traffic_src = _pmc_traffic(args.workload, "elements", n)
        result = {
            "metric": "256-bit Montgomery %s (BN254 Fr, HBM-resident batch)" % unit,
            "value": value, "unit": unit,
            "per_gpu": value / world,
            "ms_per_step": wall / args.steps * 1e3,
            "config": {"workload": "BN254 Fr 256-bit batched %s over 2^%d HBM-resident elements per GPU "
                                   "(BASELINE.json configs[1])" % (args.workload, args.log2n),
                       "elements_per_gpu": n, "limbs64": 4, "parallelism": "shard%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE; a stored rocprofv3 pass, bench.py cannot profile itself)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per * n,
                         "kernel": "k_elementwise<8,%s>" % args.workload, "bytes_per_element": bytes_per,
                         "kernel_ms": kern * 1e3},
            "verified_vs_oracle": verified,
        }
    elif args.workload == "matvec":
        result = bench_matvec(args, f, world, rank, distributed, time_steps, True, samples)
    elif args.workload == "batch_inv":
        result = bench_batch_inv(args, world, rank, time_steps, True, samples)
    elif args.workload == "matmul":
        result = bench_matmul(args, world, rank, distributed, time_steps, True, samples)

    # ---------------- secondary measurements ------------------------------------------
    if not args.no_extra:
        ksteps = max(3, min(args.steps, 10))
        # three untimed launches before each secondary measurement: the sample gathered above (and every
        # other host-side pause) lets the GPU clocks drop, and the first launches after a pause run up to
        # 6 % slow (tools/probe_sqr_timing2.py: sqr 3.18 ms right after the gather, 3.00 ms from the next series on)
        xwarm = 3
        for name, (fn, bytes_per) in ops.items():
            if name == args.workload:
                continue
            wall, kern = time_steps(fn, ksteps, xwarm)
            extra[name] = {"Gelem_per_s": n * world * ksteps / wall / 1e9, "kernel_ms": kern * 1e3,
                           "hbm_GBps_per_gpu": bytes_per * n / kern / 1e9,
                           "hbm_frac": bytes_per * n / kern / 1e9 / HBM_PEAK_GBPS}
        # the remaining elementwise rows of SURVEY.md 8a (A4, A8): one read, one write
        for name, fn in (("neg", lambda: f.neg_dev(a, out)), ("to_mont", lambda: f.to_mont_dev(a, out)),
                         ("from_mont", lambda: f.from_mont_dev(a, out)), ("double", lambda: f.double_dev(a, out)),
                         ("div_by_2", lambda: f.div_by_2_dev(a, out))):
            wall, kern = time_steps(fn, ksteps, xwarm)
            extra[name] = {"Gelem_per_s": n * world * ksteps / wall / 1e9, "kernel_ms": kern * 1e3,
                           "hbm_GBps_per_gpu": 64 * n / kern / 1e9, "hbm_frac": 64 * n / kern / 1e9 / HBM_PEAK_GBPS}
        # in place, a[i] *= b[i] (the reference's MulAssign<&Self>, crypto_bigint_const_monty.rs:337-347): the
        # same 96 algorithmic bytes per element, written to the lines just read
        wall, kern = time_steps(lambda: f.mul_dev(out, b, out), ksteps, xwarm)
        extra["mul_assign_in_place"] = {"Gelem_per_s": n * world * ksteps / wall / 1e9, "kernel_ms": kern * 1e3,
                                        "hbm_GBps_per_gpu": 96 * n / kern / 1e9, "hbm_frac": 96 * n / kern / 1e9 / HBM_PEAK_GBPS}
        # register-resident dependent chain: the VALU-bound Gmul/s ("Mul assign" shape)
        nchain, iters = 1 << 22, 256
        wall, kern = time_steps(lambda: f.mul_chain_dev(a[:nchain], b[:nchain], out[:nchain], iters), 5, 2)
        gmul = nchain * iters / kern / 1e9
        # 136 = 2*8^2 + 8 multiplies of the reference's 32-bit-limb CIOS; the kernel
        # executes 171 = 2*9^2 + 9 on unsaturated 29-bit limbs (no carry adds), DESIGN.md section 4
        extra["mul_chain"] = {"Gmul_per_s_per_gpu": gmul, "elements": nchain, "iters": iters,
                              "valu_mad_Tops": gmul * 136 / 1e3, "valu_mad_peak_Tops": VALU_MAD_PEAK_TOPS,
                              "valu_frac_vs_136_mads": gmul * 136 / 1e3 / VALU_MAD_PEAK_TOPS,
                              "valu_frac_vs_171_executed_mads": gmul * 171 / 1e3 / VALU_MAD_PEAK_TOPS}
        if args.workload != "batch_inv":
            extra["batch_inv_bls12_381"] = bench_batch_inv(args, world, rank, time_steps, False, samples)
        if args.workload != "matmul":
            extra["matmul_goldilocks"] = bench_matmul(args, world, rank, distributed, time_steps, False, samples)
        if args.workload != "matvec":
            extra["matvec_bn254"] = bench_matvec(args, f, world, rank, distributed, time_steps, False, samples)
        if world == 1:
            extra["matmul_bn254"] = bench_matmul_multilimb(f, time_steps)
        extra["spmv_bn254"] = bench_spmv(f, rank, time_steps)
        extra["pow_bn254"] = bench_pow(f, a, out, time_steps)
        if rank == 0:
            extra["drop_in_chain_bn254"] = bench_drop_in_chain(f)

    # ---------------- one process, every GPU, inside the C ABI ----------
    # The survey's own multi-GPU design: ONE process names the devices (fpgpu_set_devices), its vectors are
    # sharded / replicated handles, one host thread enqueues on every device's stream and RCCL
    # (ncclCommInitAll) gathers the mat-vec's row blocks.  Rank 0 starts it as a CHILD process (bench.py
    # --sharded-only, a time limit on it) once every rank has released its memory: the first contact with several
    # physical GPUs happens on the driver's node, and neither a failure nor a hang there may cost the bench line.
    if not args.no_extra and not args.no_sharded:
        a = b = out = None
        ops.clear()
        torch.cuda.empty_cache()
        barrier()
        if distributed:   # the last collective is behind us: the other ranks may go while rank 0 measures on
            dist.destroy_process_group()
            distributed_done = True
        if rank == 0:
            devs = [int(t) for t in args.sharded_devices.split(",")] if args.sharded_devices else list(range(world))
            extra["in_library_sharded"] = run_in_library_sharded(devs)

    # ---------------- CPU baseline (rank 0, N = 1 only) --------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(sample if elementwise else None, samples)
        if cpu.get("sample_matches_gpu") is not None:
            result["verified_vs_oracle"] = cpu["sample_matches_gpu"]
        checks = cpu.get("samples_match_gpu", {})
        if args.workload in checks:
            result["verified_vs_oracle"] = checks[args.workload]
        for key, name in (("batch_inv_bls12_381", "batch_inv"), ("matmul_goldilocks", "matmul"), ("matvec_bn254", "matvec")):
            if key in extra and name in checks:
                extra[key]["verified_vs_oracle"] = checks[name]

    if rank == 0:
        line = {
            "metric": result.get("metric"), "value": result.get("value"), "unit": result.get("unit"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": result.get("ms_per_step"), "higher_is_better": True,
            # elementwise batches: every rank owns its own 2^28 elements (weak); the inversion and the
            # matrix products split one fixed problem across the ranks (strong)
            "scaling": "weak" if elementwise else "strong",
            "vs_baseline": None,
            "dtype": {"batch_inv": "u32 limbs (12 x 32-bit per 384-bit element; 14 x 29-bit in the chain multiplier)",
                      "matmul": "u64 single-limb elements as 8 signed int8 digits (v_mfma_i32_16x16x64_i8)",
                      "matvec": "u32 limbs (8 x 32-bit per 256-bit element) as int8 operands (v_mfma_i32_16x16x64_i8)"}.get(
                          args.workload, "u32 limbs (8 x 32-bit per 256-bit element, v_mad_u64_u32)"),
            "data": "synthetic uniform field elements generated on device (SplitMix64 counter RNG)",
            "config": result.get("config"), "per_gpu": result.get("per_gpu"),
            "roofline": result.get("roofline"), "cpu_baseline": cpu,
            "verified_vs_oracle": result.get("verified_vs_oracle"),
            "distributed": {"process_group": "nccl (RCCL)" if distributed else None, "world": world,
                            "timing": "max over ranks via all_reduce(MAX)" if distributed else "single rank"},
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if distributed and not distributed_done:
        dist.destroy_process_group()


def bench_batch_inv(args, world, rank, time_steps, primary, samples):
    """BASELINE.json configs[2]: BLS12-381 Fq (384-bit precision) batch inversion over 2^24
    elements, sharded: each rank inverts its own 2^24/world slice (strong scaling)."""
    import torch
    import crypto_primitives_amd as cpa
    fq = cpa.MontyField(cpa.BLS12_381_FQ, 6)
    total = 1 << args.inv_log2
    n = total // world
    import numpy as np
    x = fq.random_dev(n, 0xB15 ^ (7 + 1000 * rank))
    out = fq.alloc(n)
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    # the larger of the two workspace sizes: level-0 prefixes stay in the multiplier's limb form (include/fpgpu.h)
    ws = torch.empty(fq.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
    # as many steps as the headline's default: five 1 ms batches right after a host-side pause sit inside the clock ramp
    # (the sqr row) and read 5-8 % slow
    steps = args.steps if primary else max(5, min(args.steps, 20))
    wall, kern = time_steps(lambda: fq.inv_dev(x, out, flags, ws), steps, max(1, args.warmup) if primary else 5)
    # property check: x * inv(x) == one wherever x != 0
    chk = fq.alloc(n)
    fq.mul_dev(x, out, chk)
    one = torch.from_numpy(fq.one_words.view(np.int64)).cuda()
    ok = bool(torch.equal(chk[flags == 0], one.expand(int((flags == 0).sum()), 6)))
    if rank == 0:
        idx = torch.arange(0, n, max(1, n // 512), device="cuda")
        samples["batch_inv"] = (x[idx].cpu().numpy().view(np.uint64), out[idx].cpu().numpy().view(np.uint64))
    # 3 Montgomery multiplications per element = 3 x (2*12^2 + 12) = 900 32-bit multiplies
    # (the kernels execute 3 x 31/32 + 0.1 of them per element, 406 multiplies each on 14 unsaturated 29-bit limbs,
    # DESIGN.md section 4)
    mads = 900.0 * n / kern / 1e12
    res = {"Ginv_per_s": total * steps / wall / 1e9, "ms_per_batch": wall / steps * 1e3,
           "elements_total": total, "bound": "valu", "mads_per_element": 900,
           "valu_mad_Tops_per_gpu": mads, "valu_frac": mads / VALU_MAD_PEAK_TOPS,
           "algorithmic_GBps_per_gpu": 96 * n / kern / 1e9, "hbm_frac": 96 * n / kern / 1e9 / HBM_PEAK_GBPS,
           "x_times_inv_is_one": ok,
           "workspace": "fpgpu_batch_inv_fast_workspace_bytes (level-0 prefixes as limbs): %d bytes; minimum %d" % (
               fq.inv_workspace_bytes(n, fast=True), fq.inv_workspace_bytes(n))}
    if primary:
        traffic, traffic_src = _pmc_traffic("batch_inv", "elements", n)
        res = {"metric": "BLS12-381 Fq batch inversion Ginv/s", "value": res["Ginv_per_s"], "unit": "Ginv/s",
               "per_gpu": res["Ginv_per_s"] / world, "ms_per_step": res["ms_per_batch"],
               "config": {"workload": "BLS12-381 Fq 384-bit batch inversion over 2^%d elements "
                                      "(BASELINE.json configs[2])" % args.inv_log2, "parallelism": "shard%d" % world},
               "roofline": {"bound": "valu", "achieved": mads, "peak": VALU_MAD_PEAK_TOPS,
                            "unit": "T 32-bit multiply-adds/s (900 per element: 3 Montgomery multiplications of 2*12^2+12)",
                            "frac": mads / VALU_MAD_PEAK_TOPS, "traffic": traffic, "traffic_source": traffic_src,
                            "traffic_unit": "HBM bytes per batch, all k_binv_* launches (PMC)",
                            "algorithmic_bytes_per_launch": 96 * n, "hbm_frac_of_8TBps": res["hbm_frac"],
                            "kernel": "k_binv_up / k_binv_top_chain / k_binv_down <12>", "kernel_ms": kern * 1e3},
               "verified_vs_oracle": ok}
    return res


def bench_matmul(args, world, rank, distributed, time_steps, primary, samples):
    """BASELINE.json configs[4]: MontyField 64-bit single-limb prime (Goldilocks) 8192^2 mat-mul.
    A and C are row-sharded across ranks, B is replicated (strong scaling).  The product stays
    row-sharded, as BASELINE.json's north_star has it (an all-gather only reassembles mat-vec outputs);
    for N > 1 the time with the 512 MiB all-gather of C added is reported beside it."""
    import torch
    import torch.distributed as dist
    import crypto_primitives_amd as cpa
    import numpy as np
    from crypto_primitives_amd import _lib
    g = cpa.MontyField(cpa.GOLDILOCKS, 1)
    dim = args.matmul_dim
    rows = dim // world
    A = g.random_dev(rows * dim, 0x60 ^ (13 + 1000 * rank))
    B = g.random_dev(dim * dim, 0x61)
    C_local = g.alloc(rows * dim)
    C = g.alloc(dim * dim) if distributed else C_local

    ws = torch.empty(g.matmul_workspace_bytes(rows, dim, dim), dtype=torch.uint8, device="cuda")

    def step():
        g.matmul_dev(A, B, C_local, rows, dim, dim, workspace=ws)

    def step_gathered():
        g.matmul_dev(A, B, C_local, rows, dim, dim, workspace=ws)
        dist.all_gather_into_tensor(C, C_local)

    steps = max(2, min(args.steps, 5)) if not primary else args.steps
    wall, kern = time_steps(step, steps, max(1, args.warmup) if primary else 2)
    if rank == 0:   # twelve entries of C with their rows of A and columns of B, for the oracle in the CPU leg
        torch.cuda.synchronize()
        rs, cs = [0, rows // 2 + 1, rows - 1], [0, 1, dim // 2 + 3, dim - 1]
        Bm, Cm = B.view(dim, dim), C_local.view(rows, dim)
        samples["matmul"] = (np.stack([A[r * dim:(r + 1) * dim].cpu().numpy().view(np.uint64).reshape(-1) for r in rs]),
                             np.stack([Bm[:, c].contiguous().cpu().numpy().view(np.uint64).reshape(-1) for c in cs]),
                             np.array([[int(Cm[r, c].item()) & 0xFFFFFFFFFFFFFFFF for c in cs] for r in rs], dtype=np.uint64), dim)
    gathered_ms = None
    if distributed:
        wall_g, _ = time_steps(step_gathered, max(2, min(steps, 5)), 1)
        gathered_ms = wall_g / max(2, min(steps, 5)) * 1e3
    # the VALU kernel on the same product, timed live (tuning hook; same results)
    _lib.tuning_set("FPGPU_MATMUL_MFMA", 0)
    try:
        _, kern_valu = time_steps(step, 2, 1)
    finally:
        _lib.tuning_set("FPGPU_MATMUL_MFMA")
    terms = dim ** 3
    # int8 multiply-accumulates the matrix cores execute: 64 digit products per 64x64-bit term
    tops = 2.0 * 64.0 * rows * dim * dim / kern / 1e12
    res = {"Gterms_per_s": terms * steps / wall / 1e9, "ms_per_matmul": wall / steps * 1e3, "dim": dim,
           "int8_mfma_Tops_per_gpu": tops, "int8_mfma_peak_Tops": MFMA_I8_PEAK_TOPS,
           "mfma_frac": tops / MFMA_I8_PEAK_TOPS,
           "kernels": "k_mf_digitize x2 + k_matmul_mfma_1limb (v_mfma_i32_16x16x64_i8)",
           "int8_peak_note": MFMA_I8_PEAK_NOTE,
           "ms_per_matmul_with_allgather_of_C": gathered_ms,
           "valu_kernel_ms_same_run": kern_valu * 1e3}
    if primary:
        extra_fields = {"ms_per_step_with_allgather_of_C": gathered_ms}
        # HBM bytes per launch of the main kernel from the stored PMC passes (N = 1 shape)
        traffic, traffic_src = _pmc_traffic("matmul", "dim", dim) if world == 1 else (None, "not collected for sharded shapes")
        res = {"metric": "64-bit Fp mat-mul Gterms/s (Goldilocks, 8192^2)", "value": res["Gterms_per_s"],
               "unit": "Gterms/s", "per_gpu": res["Gterms_per_s"] / world, "ms_per_step": res["ms_per_matmul"],
               "config": {"workload": "Goldilocks 64-bit %d x %d x %d mat-mul (BASELINE.json configs[4])" % (dim, dim, dim),
                          "parallelism": "rowshard%d (C stays row-sharded)" % world, **extra_fields},
               "roofline": {"bound": "mfma", "achieved": tops, "peak": MFMA_I8_PEAK_TOPS,
                            "unit": "TOP/s (int8 multiply-accumulates x 2; 64 per 64x64-bit term)",
                            "frac": tops / MFMA_I8_PEAK_TOPS, "peak_note": MFMA_I8_PEAK_NOTE, "traffic": traffic,
                            "traffic_source": traffic_src,
                            "traffic_unit": "HBM bytes per launch of k_matmul_mfma_1limb (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                            "kernel": "k_matmul_mfma_1limb"},
               "verified_vs_oracle": None}
    return res


def bench_matmul_multilimb(f, time_steps, dim=2048):
    """Multi-limb mat-mul (BN254 Fr, dim^3; not a BASELINE configuration) on the int8 matrix cores as n64 x n64
    single-limb passes (fp_matmul_mfma_limbs.cuh), with the VALU kernel on the same product timed in the same run and
    the two results compared bit for bit (the VALU kernel is the one the parity tests hold to the oracle at this size)."""
    import torch
    from crypto_primitives_amd import _lib
    A, B = f.random_dev(dim * dim, 0x71), f.random_dev(dim * dim, 0x72)
    C, C2 = f.alloc(dim * dim), f.alloc(dim * dim)
    nb = f.matmul_workspace_bytes(dim, dim, dim)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
    wall, kern = time_steps(lambda: f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws), 3, 2)
    _lib.tuning_set("FPGPU_MATMUL_MFMA", 0)
    try:
        _, kern_valu = time_steps(lambda: f.matmul_dev(A, B, C2, dim, dim, dim), 1, 1)
    finally:
        _lib.tuning_set("FPGPU_MATMUL_MFMA")
    torch.cuda.synchronize()
    n64 = f.n64
    tops = 2.0 * 64.0 * n64 * n64 * dim ** 3 / kern / 1e12
    return {"Gterms_per_s": dim ** 3 / kern / 1e9, "ms_per_matmul": kern * 1e3, "dim": dim, "workspace_MiB": nb / 2 ** 20,
            "int8_mfma_Tops_per_gpu": tops, "int8_mfma_peak_Tops": MFMA_I8_PEAK_TOPS, "mfma_frac": tops / MFMA_I8_PEAK_TOPS,
            "kernels": "k_mfl_digitize x2 + k_matmul_mfma_limbs (%d x %d passes of the v_mfma_i32_16x16x64_i8 loop)" % (n64, n64),
            "valu_kernel_ms_same_run": kern_valu * 1e3, "same_bits_as_valu_kernel": bool(torch.equal(C, C2))}


def bench_matvec(args, f, world, rank, distributed, time_steps, primary, samples):
    """BASELINE.json configs[3]: 256-bit Fp dense mat-vec, rows sharded across ranks,
    y reassembled with an RCCL all-gather (the only collective on the path)."""
    import torch
    import torch.distributed as dist
    dim = 1 << args.matvec_log2
    rows = dim // world
    A = f.random_dev(rows * dim, 0xA ^ (11 + 1000 * rank))
    x = f.random_dev(dim, 0xC0FFEE)           # same x on every rank (replicated operand)
    y_local = f.alloc(rows)
    y = f.alloc(dim) if distributed else y_local

    ws = torch.empty(f.matvec_workspace_bytes(rows, dim), dtype=torch.uint8, device="cuda")

    def step():
        f.matvec_dev(A, rows, dim, x, y_local, workspace=ws)
        if distributed:
            dist.all_gather_into_tensor(y, y_local)

    steps = args.steps if primary else 5
    wall, kern = time_steps(step, steps, max(1, args.warmup) if primary else 3)
    if rank == 0:   # three rows of A, x and their outputs for the oracle in the CPU leg
        import numpy as np
        torch.cuda.synchronize()
        rs = [0, rows // 2 + 1, rows - 1]
        samples["matvec"] = (np.stack([A[r * dim:(r + 1) * dim].cpu().numpy().view(np.uint64) for r in rs]),
                             x.cpu().numpy().view(np.uint64), y_local[rs].cpu().numpy().view(np.uint64), dim)
    elems = dim * dim
    res = {"matrix_elems_per_s": elems * steps / wall, "ms_per_matvec": wall / steps * 1e3,
           "dim": dim, "hbm_GBps_per_gpu": 32.0 * (rows * dim + dim + rows) / kern / 1e9,
           "hbm_frac": 32.0 * (rows * dim + dim + rows) / kern / 1e9 / HBM_PEAK_GBPS}
    if primary:
        res = {"metric": "256-bit Fp mat-vec matrix elems/s", "value": res["matrix_elems_per_s"],
               "unit": "elems/s", "per_gpu": res["matrix_elems_per_s"] / world,
               "ms_per_step": res["ms_per_matvec"],
               "config": {"workload": "BN254 Fr 2^%d x 2^%d dense mat-vec, row-sharded "
                                      "(BASELINE.json configs[3])" % (args.matvec_log2, args.matvec_log2),
                          "parallelism": "rowshard%d+allgather" % world},
               "roofline": {"bound": "hbm", "achieved": res["hbm_GBps_per_gpu"], "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": res["hbm_frac"], "traffic": _pmc_traffic("matvec", "rows", rows)[0],
                            "traffic_source": _pmc_traffic("matvec", "rows", rows)[1],
                            "kernel": "k_matvec_mfma<8,4> (+ k_mv_expand, k_mv_finish)"},
               "verified_vs_oracle": None}
    return res


def run_in_library_sharded(devices, limit_s=300):
    """bench.py --sharded-only as a child process: its one JSON object, or what went wrong."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-only", "--sharded-devices", ",".join(str(d) for d in devices)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                            "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=env)
    except subprocess.TimeoutExpired:
        return {"devices": list(devices), "error": "no result within %d s (child process killed)" % limit_s}
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith("{") and "in_library_sharded" in ln:
            return json.loads(ln)["in_library_sharded"]
    return {"devices": list(devices), "error": "child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}


def bench_in_library_sharded(devices, steps=5, warmup=2):
    """SURVEY.md 8e as written: a single process drives `devices` through the C ABI -- fpgpu_set_devices,
    FPGPU_PLACE_SHARDED / FPGPU_PLACE_REPLICATED handles, one stream per list entry, no torch.distributed.
    Elementwise mul (2^26 BN254 elements per entry, weak) and batch inversion (2^24 BLS12-381 elements in total)
    shard with no collective; the 2^15 x 2^15 mat-vec shards rows (the reference's rayon axis,
    src/matrix.rs:170-184), x replicated, and is timed twice: y left row-sharded, and y replicated, i.e. with the
    all-gather of the row blocks (ncclAllGather over the communicators of ncclCommInitAll; event-ordered peer
    copies when the list repeats a device) -- the difference is the collective's cost.  Wall time on the host
    around `steps` calls, every device synchronised at both ends."""
    import numpy as np
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import vec as V
    g = len(devices)
    res = {"devices": list(devices), "process": "one process, one host thread, one stream per list entry",
           "transport": "RCCL ncclAllGather (ncclCommInitAll)" if len(set(devices)) == g and g > 1 else
                        ("none (one entry)" if g == 1 else "event-ordered peer copies (the list repeats a device; RCCL refuses that)")}

    def timed(fn, sync):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        return (time.perf_counter() - t0) / steps

    V.set_devices(devices)
    try:
        f = cpa.MontyField(cpa.BN254_FR, 4)
        n = (1 << 26) * g
        va = V.DeviceVector.alloc(f, n, V.PLACE_SHARDED).fill_random(0x51)
        vb = V.DeviceVector.alloc(f, n, V.PLACE_SHARDED).fill_random(0x52)
        vo = V.DeviceVector.alloc(f, n, V.PLACE_SHARDED)
        dt = timed(lambda: f.mul_vec(va, vb, vo), vo.sync)
        res["mul_bn254"] = {"elements_total": n, "ms": dt * 1e3, "Gmul_per_s": n / dt / 1e9,
                            "hbm_frac_per_gpu": 96.0 * n / dt / 1e9 / HBM_PEAK_GBPS / len(set(devices))}
        for v in (va, vb, vo):
            v.free()

        fq = cpa.MontyField(cpa.BLS12_381_FQ, 6)
        n = 1 << 24
        vx = V.DeviceVector.alloc(fq, n, V.PLACE_SHARDED).fill_random(0x53)
        vi = V.DeviceVector.alloc(fq, n, V.PLACE_SHARDED)
        vc = V.DeviceVector.alloc(fq, n, V.PLACE_SHARDED)
        dt = timed(lambda: fq.inv_vec(vx, vi, want_flags=False), vi.sync)
        fq.mul_vec(vx, vi, vc)
        prod = vc.download(0, 4096)
        res["batch_inv_bls12_381"] = {"elements_total": n, "ms": dt * 1e3, "Ginv_per_s": n / dt / 1e9,
                                      "x_times_inv_is_one_on_sample": bool(np.array_equal(prod, np.broadcast_to(fq.one_words, prod.shape)))}
        for v in (vx, vi, vc):
            v.free()

        dim = 1 << 15
        A = V.DeviceVector.alloc_rows(f, dim, dim, V.PLACE_SHARDED).fill_random(0x54)
        x = V.DeviceVector.alloc(f, dim, V.PLACE_REPLICATED).fill_random(0x55)
        ys = V.DeviceVector.alloc(f, dim, V.PLACE_SHARDED)
        yr = V.DeviceVector.alloc(f, dim, V.PLACE_REPLICATED)
        t_sh = timed(lambda: f.matvec_vec(A, dim, dim, x, ys), ys.sync)
        t_rep = timed(lambda: f.matvec_vec(A, dim, dim, x, yr), yr.sync)
        same = bool(np.array_equal(ys.download(), yr.download()))
        res["matvec_bn254"] = {"dim": dim, "ms_y_row_sharded": t_sh * 1e3, "ms_y_replicated_with_allgather": t_rep * 1e3,
                               "allgather_ms": max(0.0, t_rep - t_sh) * 1e3,   # a difference of two timings: never reported below zero
                               "allgather_bytes_per_entry": 32 * dim // g,
                               "matrix_elems_per_s": dim * dim / t_rep, "gathered_equals_sharded": same,
                               "hbm_GBps_total": 32.0 * (dim * dim + dim + dim) / t_sh / 1e9}
        for v in (A, x, ys, yr):
            v.free()
    finally:
        V.set_devices([])
    return res


def bench_spmv(f, rank, time_steps):
    """SparseMatrix mat-vec, one replica per rank: 2^20 rows x 32 cells per row over
    2^20 columns, BN254 Fr; uniform random columns, so x is gathered through L2 / Infinity Cache."""
    import torch
    rows = cols = 1 << 20
    density = 32
    vals = f.random_dev(rows * density, 0x5A ^ (17 + 1000 * rank))
    x = f.random_dev(cols, 0x5B)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    col = torch.randint(0, cols, (rows * density,), dtype=torch.int32, device="cuda", generator=g)
    y = f.alloc(rows)
    wall, kern = time_steps(lambda: f.spmv_dev(vals, col, rows, cols, density, x, y), 5, 1)
    bytes_alg = 36.0 * rows * density + 32.0 * (cols + rows)   # value + index per cell, x and y once
    return {"cells_per_s": rows * density / kern, "ms": kern * 1e3, "rows": rows, "density": density,
            "hbm_GBps_per_gpu": bytes_alg / kern / 1e9, "hbm_frac": bytes_alg / kern / 1e9 / HBM_PEAK_GBPS}


def bench_pow(f, a, out, time_steps):
    """Batched pow, one replica per rank: x^(p-2) over 2^22 BN254 elements -- the Fermat
    inverse, a dense 254-bit exponent: 253 squarings + 67 multiplications on the sliding-window ladder."""
    n = 1 << 22
    e = f.modulus_int - 2
    wall, kern = time_steps(lambda: f.pow_dev(a[:n], e, out[:n]), 3, 1)
    return {"Gpow_per_s": n / kern / 1e9, "ms": kern * 1e3, "elements": n, "exponent": "p - 2",
            "Gmul_equiv_per_s_binary_ladder_count": n * (e.bit_length() - 2 + bin(e).count("1")) / kern / 1e9}


def bench_drop_in_chain(f):
    """What the boundary costs a host caller that chains operations (t = a*b; u = t + a; v = u^2 over 2^22 BN254
    elements, host arrays in, host array out): the host-buffer calls move every operand of every call over
    PCIe (5 uploads, 3 downloads); the handle API uploads a and b once, keeps t and u in HBM and downloads v
    once; the third line is the same chain on vectors that are already resident (no PCIe at all)."""
    import numpy as np
    import crypto_primitives_amd as cpa
    n = 1 << 22
    rng = np.random.default_rng(5)
    ha = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    hb = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    ha[:, 3] &= (1 << 60) - 1
    hb[:, 3] &= (1 << 60) - 1

    def host_calls():
        return f.sqr(f.add(f.mul(ha, hb), ha))

    V = cpa.DeviceVector
    vt, vu, vv = V.alloc(f, n), V.alloc(f, n), V.alloc(f, n)
    va, vb = V.alloc(f, n), V.alloc(f, n)

    def handles():
        va.upload(ha)
        vb.upload(hb)
        f.mul_vec(va, vb, vt)
        f.add_vec(vt, va, vu)
        f.sqr_vec(vu, vv)
        return vv.download()

    def resident():
        f.mul_vec(va, vb, vt)
        f.add_vec(vt, va, vu)
        f.sqr_vec(vu, vv)
        vv.sync()

    out = {}
    ref = None
    for name, fn, reps in (("host_buffer_calls", host_calls, 3), ("handles_upload_chain_download", handles, 3),
                           ("handles_resident", resident, 20)):
        # steady state: the first two or three host calls of a process run at half speed (first use of the
        # staging lanes' pinned buffers and threads; tools/probe_bench_chain.py: 14, 14, 7.6, 7.6 ms for mul)
        for _ in range(3):
            r = fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"ms": dt * 1e3, "Gelem_per_s": n / dt / 1e9}
        if r is not None:
            same = True if ref is None else bool(np.array_equal(r, ref))
            ref = r if ref is None else ref
            out[name]["same_result"] = same
    out["elements"] = n
    out["ops"] = "t = a*b; u = t + a; v = u^2"
    return out


def _latest_profile(name):
    """profiles/r<NN>_<name> of the newest round (the summaries are kept per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + name)))
    return found[-1] if found else os.path.join(ROOT, "profiles", name)


def kernel_sources_sha256():
    """Digest of every kernel source the library is built from.  A stored PMC pass carries the digest of
    the sources it measured; bench.py only quotes it while the sources are still those."""
    import glob
    import hashlib
    h = hashlib.sha256()
    host_only = ("fpgpu_vec.hip", "fpgpu_stage.hip", "fpgpu_internal.hpp")   # handles, copies: no kernel, no launch geometry
    for path in sorted(glob.glob(os.path.join(ROOT, "crypto_primitives_amd", "csrc", "*"))):
        if path.endswith((".cuh", ".hip", ".hpp", ".inc")) and os.path.basename(path) not in host_only:
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def _pmc_traffic(workload, key, value):
    """(HBM bytes per launch, provenance) from the PMC passes kept under profiles/ (bench.py cannot run the
    profiler on itself).  Fails closed: no figure when the stored pass measured another problem size
    or other kernel sources than the ones in the tree."""
    path = _latest_profile("pmc_traffic_%s.json" % workload)
    try:
        with open(path) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        return None, "no stored PMC pass for this workload"
    name = os.path.relpath(path, ROOT)
    if pmc.get(key) != value:
        return None, "%s measured %s = %s, not %s" % (name, key, pmc.get(key), value)
    if pmc.get("kernel_sources_sha256") != kernel_sources_sha256():
        return None, "%s is stale: the kernel sources changed after it was collected (commit %s)" % (name, pmc.get("commit"))
    return pmc.get("traffic_bytes_per_launch"), "%s (collected at commit %s, same kernel sources as this tree)" % (name, pmc.get("commit"))


def cpu_baseline(sample=None, samples=None):
    """The CPU restatement of the reference's path (oracle/, "port"), timed on this box's host
    cores: 2^20-element BN254 Fr batches (BASELINE.json configs[0] shape), repeated for a
    bounded ~10 s, all OpenMP threads the box grants.  This leg is also where the GPU's output
    sample (workload, a, b, out) is checked against the oracle - the only place bench.py uses it."""
    import numpy as np
    from oracle.fp_oracle import Oracle
    import crypto_primitives_amd as cpa
    o = Oracle(cpa.BN254_FR, 4)
    matches = None
    if sample is not None:
        op, sa, sb, sout = sample
        exp = {"mul": o.mul, "add": o.add, "sub": o.sub}.get(op)
        exp = exp(sa, sb) if exp else o.sqr(sa)
        matches = bool(np.array_equal(sout, exp))
    checks = {}
    for name, smp in (samples or {}).items():
        if name == "matvec":
            rowsA, x, y, dim = smp
            checks[name] = all(np.array_equal(o.matvec(rowsA[i], 1, dim, x, threads=0).reshape(-1), y[i].reshape(-1))
                               for i in range(len(rowsA)))
        elif name == "matmul":
            rowsA, colsB, entries, dim = smp    # C[r][c] = sum_l mul(A[r][l], B[l][c]): the fold of mul then add
            og = Oracle(cpa.GOLDILOCKS, 1)
            checks[name] = all(int(og.sum(og.mul(rowsA[i].reshape(-1, 1), colsB[j].reshape(-1, 1)))[0]) == int(entries[i, j])
                               for i in range(len(rowsA)) for j in range(len(colsB)))
        elif name == "batch_inv":
            xs, inv = smp
            oq = Oracle(cpa.BLS12_381_FQ, 6)
            checks[name] = bool(np.array_equal(oq.batch_inv(xs)[0], inv))
    n = 1 << 20
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= (1 << 60) - 1
    b[:, 3] &= (1 << 60) - 1
    # the box's CPU share (OMP_NUM_THREADS is 16 for one GPU); set explicitly because
    # libraries loaded earlier in this process may have lowered OpenMP's default
    avail = len(os.sched_getaffinity(0))
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS") or avail), avail))
    o.mul(a, b, threads=threads)  # warm up the thread pool
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 8.0:
        o.mul(a, b, threads=threads)
        reps += 1
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    o.mul(a[: n // 8], b[: n // 8], threads=1)
    single = (n // 8) / (time.perf_counter() - t1) / 1e9
    # configs[0] also names add: the same 2^20-element batch through add_mod, ~2 s
    o.add(a, b, threads=threads)
    areps, t2 = 0, time.perf_counter()
    while time.perf_counter() - t2 < 2.0:
        o.add(a, b, threads=threads)
        areps += 1
    add_rate = areps * n / (time.perf_counter() - t2) / 1e9
    # the reference's own bench shape (benches/field_ops_bench_common.rs): 10000 operations on one element,
    # four seeds, eight groups, single core; on its 120-bit bench modulus and on BN254 Fr
    from oracle.fp_oracle import bench_reference_shape
    p120 = 0x00000000000000000000000000000000860995AE68FC80E1B1BD1E39D54B33  # benches/const_monty_field_benches.rs:9-13
    ref_shape = {}
    for label, mod in (("bench_120bit_prime", p120), ("bn254_fr", cpa.BN254_FR)):
        per_seed = bench_reference_shape(mod, 4, 10000)
        groups = next(iter(per_seed.values())).keys()
        ref_shape[label] = {g: round(sum(v[g] for v in per_seed.values()) / len(per_seed) / 1e4, 2) for g in groups}
    return {"value": reps * n / dt / 1e9, "unit": "Gmul/s", "cores": threads, "kind": "port",
            "reference_bench_shape_ns_per_op": ref_shape,
            "sample": "%d passes of a 2^20-element BN254 Fr mul batch (C restatement of monty_mul, "
                      "-O3, OpenMP static chunks); single-thread rate %.4f Gmul/s" % (reps, single),
            "single_thread_Gmul_per_s": single, "add_Gadd_per_s": add_rate, "sample_matches_gpu": matches,
            "samples_match_gpu": checks}


if __name__ == "__main__":
    main()
