#!/usr/bin/env python3
"""bench.py -- headline benchmark of the distance + top-k hot path (BASELINE.json).

A "step" is one pass of the hot path over one batch: the brute-force k-NN search of M
queries against the N-row fp32 base (config C2: N=1M, d=128, M=10k, k=10, L2), inputs
resident in HBM.  With --gpus G > 1 (launched by torch.distributed.run, one rank per GPU) the
base is sharded by contiguous row ranges (strong scaling: total N and total queries fixed):
pure row sharding, as BASELINE's north_star words it -- rank r holds rows [r*ceil(N/G), ...), every
rank scans its rows for ALL queries, the per-shard top-k are exchanged with ONE RCCL all-gather and
merged, all behind the C ABI (expann_sharded_*, csrc/expann_sharded.hip; torch.distributed only
hands out the RCCL unique id and provides the barriers around the timed region).
--row-shards R < G selects the hybrid grid of expann_amd/sharded.py instead (R row shards x G/R
query groups, exchanged through torch.distributed), --exchange torch the same collectives through
torch.distributed.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, timed with HIP events on its stream inside the library), `cpu_baseline` (the CPU
oracle timed on this box's host cores on a bounded sample; N=1 only) and the bench's own proof of
its claim: `recall_measured`, `verified_queries`, `bit_exact` (the GPU result of the timed step
against the oracle on a sample of the step's queries; a mismatch exits non-zero).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32-input MFMA peak (155 measured)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
MFMA_I8_PEAK_TOPS = 5033.0     # int8 MFMA = 2x the bf16 rate per clock (~2.5 PF dense bf16):
                               # 2048 ops/clk/SIMD x 1024 SIMDs x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    # (--rows / --dim / --queries: spellings that survive torch.distributed.run's own option parser,
    # which rejects --n, --d and --m as ambiguous abbreviations of its flags)
    ap.add_argument("--n", "--rows", type=int, default=1_000_000)
    ap.add_argument("--d", "--dim", type=int, default=128)
    ap.add_argument("--m", "--queries", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--dtype", default="f32", choices=["f32", "i8", "u8"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "c4"],
                    help="preset: c2 = BASELINE configs[1] (default, the headline metric); "
                         "c3 = configs[2] 10Mxd128 fp32, k=100 (sets n / k; with --gpus G the rows are "
                         "sharded G ways as the config asks, one GPU holds all 10 M); "
                         "c5 = configs[4] int8 IP 10Mxd768 (sets n/d/dtype/metric); "
                         "c4 = configs[3] graph search recall sweep (SIFT-like stand-in, --rows rows, "
                         "M=60 / M0=120 / ef_construction=480, built by the batched GPU builder)")
    ap.add_argument("--query-tile", type=int, default=0)
    ap.add_argument("--row-shards", type=int, default=0,
                    help="row shards of the rank grid (default: --gpus = pure row sharding, every rank "
                         "searches every query; R < --gpus: R row shards x gpus/R query groups)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "native", "torch"],
                    help="native: RCCL all-gather + merge behind the C ABI (expann_sharded_*; default for "
                         "pure row sharding); torch: the same exchange through torch.distributed "
                         "(always for the hybrid grid and the one-GPU rehearsal)")
    ap.add_argument("--exchange-pattern", default="auto", choices=["auto", "allgather", "slices"],
                    help="sharded handle (expann_sharded_*): slices (= auto) -- all-to-all of query slices, each "
                         "rank merges m/G queries; allgather -- round 2's one all-gather of whole chunks")
    ap.add_argument("--verify-queries", type=int, default=0,
                    help="queries of the step checked against the CPU oracle when no cpu_baseline leg "
                         "runs (G > 1 or --no-cpu-baseline); default 8, 0 with --no-verify")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed loop rank 0 checks the sharded result bit for bit against an "
                         "unsharded search of the whole base (f32 only)")
    ap.add_argument("--debug", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sample-ratio", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sample-frac", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--xcd-tolerance", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--scan-chunks", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sync-search", action="store_true",
                    help="wait for every search before enqueuing what follows (default: deferred check, "
                         "one expann_sync at the end of the timed steps)")
    ap.add_argument("--sample-run", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sift-like", action="store_true",
                    help="f32 rows and queries hold integers 0..255 (SURVEY 8d's SIFT stand-in)")
    ap.add_argument("--clustered", type=int, default=0,
                    help="synthetic base = this many Gaussian clusters stored contiguously (robustness "
                         "probe; default 0 = iid rows)")
    ap.add_argument("--scan-kernel", type=int, default=0,
                    help="0 auto, 1 direct, 2 GEMM form fp32/int8 MFMA, 3 GEMM form bf16x3")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="rough budget of the CPU baseline leg")
    a = ap.parse_args()
    if a.workload == "c3":
        if a.n == 1_000_000:
            a.n = 10_000_000
        if a.k == 10:
            a.k = 100
    if a.workload == "c5":
        a.dtype, a.metric = "i8", "ip"
        if a.n == 1_000_000 and a.d == 128:
            a.n, a.d = 10_000_000, 768
    return a


def cpu_baseline(base_host, queries_host, k, budget_s, metric_name="METRIC_L2_F32"):
    """Time the CPU oracle (kind 'port': oracle/expann_oracle.c, a restatement of
    src/brute_force_engine.h:28-46 + src/distance.h:136-147) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_ctypes as oc
    lib = None
    try:  # a -march=native build for the host we are on; fall back to the shipped x86-64-v3 one
        out_dir = os.path.join(ROOT, "gpurun_out", "oracle_native")
        lib = oc.lib(oc.build(arch_flags=["-march=native"], out_dir=out_dir))
        flags = "-O3 -march=native"
    except Exception:
        lib = oc.lib()
        flags = "-O3 -march=x86-64-v3 (prebuilt)"
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2 CPU quota of this box (the GPU box gives a share of the host's cores)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except Exception:
        pass
    # 1 thread: the reference's execution model (src/basic_bench.h:83-84)
    t0 = time.perf_counter()
    oc.brute_force(base_host, queries_host[:2], k, getattr(oc, metric_name), 1, _lib_override=lib)
    t1 = (time.perf_counter() - t0) / 2
    n1 = max(2, min(32, int(budget_s * 0.3 / max(t1, 1e-6))))
    t0 = time.perf_counter()
    oc.brute_force(base_host, queries_host[:n1], k, getattr(oc, metric_name), 1, _lib_override=lib)
    qps1 = n1 / (time.perf_counter() - t0)
    # all cores, queries partitioned over threads (extension).  The scan is DRAM
    # bound, so more threads than memory channels can be slower: try a few pool sizes and
    # report the best one (cores = the threads actually used for the reported value).
    best = (0.0, 1, 0)
    checked = None      # oracle (ids, dists) of the longest prefix of the step's queries it scanned
    cands = sorted({min(avail, c) for c in (16, 32, 64, avail)})
    per = budget_s * 0.7 / len(cands)
    for c in cands:
        nq = int(min(len(queries_host), max(c, per * qps1 * min(c, 16) * 0.5)))
        nq = max(c, (nq // c) * c)
        t0 = time.perf_counter()
        res = oc.brute_force(base_host, queries_host[:nq], k, getattr(oc, metric_name), c, _lib_override=lib)
        q = nq / (time.perf_counter() - t0)
        if q > best[0]:
            best = (q, c, nq)
        if checked is None or nq > checked[0].shape[0]:
            checked = res
    qps_all, cores, nall = best
    return {"value": round(qps_all, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{nall} of the step's queries x all {base_host.shape[0]} rows on {cores} "
                      f"threads ({flags}); 1 thread: {qps1:.2f} queries/s over {n1} queries",
            "single_thread_value": round(qps1, 2)}, checked


def oracle_check(oracle_ids, oracle_d, gpu_ids, gpu_d):
    """The bench's proof of its claim (the reference computes recall inside its harness,
    src/basic_bench.h:106-121,143): GPU ids / distance bits of the first rows of the step's result
    against the oracle's for the same queries.  -> (recall, n_queries, bit_exact)"""
    import numpy as np
    nq, k = oracle_ids.shape
    g_ids = np.ascontiguousarray(gpu_ids[:nq]).view(np.uint64)
    g_d = np.ascontiguousarray(gpu_d[:nq])
    found = sum(len(np.intersect1d(g_ids[i], oracle_ids[i])) for i in range(nq))
    valid = int((oracle_ids != np.uint64(2 ** 64 - 1)).sum())
    bit_exact = bool(np.array_equal(g_ids, oracle_ids)) and \
        bool(np.array_equal(g_d.view(np.uint32), oracle_d.view(np.uint32)))
    return found / max(1, valid), nq, bit_exact


def profiled_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed rocprofv3 PMC summary
    (profiles/rNN_summary.json, made by profiles/run_rocprof.sh + summarize.py on the same
    default command).  PMC counters cannot be read from inside the timed process, so this is
    the per-launch figure of that profile run; None when no profile matches the kernel."""
    import glob
    import re

    def norm(name):
        mm = re.match(r"(?:void )?(?:expann::)?(\w+?)(?:_kernel)?<([^>]*)>", name or "")
        if not mm:
            return None
        args = []
        for x in mm.group(2).split(","):
            x = x.strip()
            args.append({"L2": 0, "false": 0, "IP": 1, "true": 1}.get(x, x))
        return (mm.group(1), tuple(str(x) for x in args[:2]))   # (dimension, variant); further template arguments are build switches

    want = norm(kernel_name)
    if not want:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            summ = json.load(open(path))
        except Exception:
            continue
        for k, e in summ.get("runs", {}).get("c2", {}).get("kernels", {}).items():
            if norm(k) == want and "hbm_traffic_bytes_per_launch" in e:
                return e["hbm_traffic_bytes_per_launch"], os.path.basename(path)
    return None, None


def bench_c4(a):
    """configs[3]: antitopo graph search, GPU candidate scoring + queues inside the traversal,
    recall@k sweep over ef_search in the reference's own configuration (src/bench_runner.h:133-162:
    M = 60, M0 = 120, ef_construction = 480, ef_search = k x {1..6}, both compression modes).
    SIFT1M is not available offline: SIFT-like synthetic rows.  The graph is built by the
    batched GPU builder (csrc/graph_build.hpp; the first rows by the serial host restatement); the
    timed part is the batched query launch.  The GPU walk is checked against the oracle's walk of the
    same index file on a sample of the queries (ids and distance bits)."""
    import subprocess
    import numpy as np
    tool = os.path.join(ROOT, "expann_amd", "host", "expann_graph_tool")
    import atexit
    import shutil
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="expann_c4_")   # index + query + result files: scratch only
    atexit.register(shutil.rmtree, out_dir, ignore_errors=True)
    n = a.n
    M, efc = 60, 480
    idx, qf, rf = (os.path.join(out_dir, x) for x in ("c4.index", "c4.queries", "c4.results"))
    efs = [a.k * mult for mult in (1, 2, 3, 4, 5, 6)]       # src/bench_runner.h:134
    cmd = [tool, "--n", str(n), "--m", str(a.m), "--d", str(a.d), "--k", str(a.k), "--M", str(M),
           "--ef_construction", str(efc), "--data", "sift", "--index", idx, "--queries", qf, "--results", rf,
           "--ef", ",".join(map(str, efs)), "--batched", "1024"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        print(res.stderr[-2000:], file=sys.stderr)
        sys.exit(res.returncode)
    lines = [json.loads(x) for x in res.stdout.strip().splitlines()]
    build = lines[0]
    sweep = [x for x in lines if x["phase"] == "query"]
    fp32 = [x for x in sweep if x["use_compression"] == 0]
    best = max(fp32, key=lambda x: x["recall"])
    gather_bytes = best["distcomps_per_query"] * a.d * 4 * a.m
    ach = gather_bytes / (best["kernel_ms"] * 1e-3) / 1e9
    out = {"metric": f"queries/sec at recall@{a.k} (graph search sweep), {n}xd{a.d} fp32 SIFT-like, k={a.k}",
           "value": round(1e9 / best["time_per_query_ns"], 1), "unit": "queries/s", "n_gpus": 1,
           "steps": 1, "warmup": 1, "ms_per_step": round(best["time_per_query_ns"] * a.m / 1e6, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"antitopo_engine graph search N={n} d={a.d}, M={M} M0={2 * M} "
                                  f"ef_construction={efc}, {a.m} batched queries, k={a.k} "
                                  "(BASELINE configs[3], SIFT-like stand-in)",
                      "recall": best["recall"], "ef_search": best["ef_search"],
                      "build_s": round(build["time_to_build_ns"] / 1e9, 1), "builder": build.get("builder"),
                      "build_stats": {k: build.get(k) for k in ("batches", "dropped_reverse_edges", "rows_repruned")},
                      "sweep": sweep},
           "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel": f"graph_search<{a.d},fp32>", "kernel_ms": best["kernel_ms"],
                        "note": "random 512-B row gathers: latency-bound, not bandwidth-bound; "
                                "algorithmic bytes = distcomps x d x 4 (SURVEY 8d)"}}
    rc = 0
    if not a.no_cpu_baseline or not a.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ctypes as oc
        g = oc.Graph(idx)
        nq = min(200, a.m)
        q = np.fromfile(qf, dtype=np.float32).reshape(-1, a.d)[:nq]
        t1 = time.perf_counter()
        oids, od, odc = g.query_k(q, a.k, int(best["ef_search"]))
        dt = time.perf_counter() - t1
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = {"value": round(len(q) / dt, 1), "unit": "queries/s", "cores": 1,
                                   "kind": "port", "sample": f"{len(q)} queries, same index, same "
                                   f"ef_search={best['ef_search']} (oracle restatement of _query_k)"}
        if not a.no_verify:
            # results file: per (compression, ef): ids m*k u64, dists m*k f32, distcomps m u32
            per = a.m * a.k * 12 + a.m * 4
            pos = [x["ef_search"] for x in fp32].index(best["ef_search"]) * per
            raw = np.fromfile(rf, dtype=np.uint8)
            ids = raw[pos:pos + a.m * a.k * 8].view(np.uint64).reshape(a.m, a.k)[:nq]
            dd = raw[pos + a.m * a.k * 8:pos + a.m * a.k * 12].view(np.float32).reshape(a.m, a.k)[:nq]
            exact = bool(np.array_equal(ids, oids)) and bool(np.array_equal(dd.view(np.uint32), od.view(np.uint32)))
            out["verified_queries"] = nq
            out["bit_exact"] = exact
            out["recall_measured"] = best["recall"]
            out["verified_against"] = ("oracle/expann_oracle_graph.c (CPU restatement of src/antitopo_engine.h:853-928) "
                                       "walking the same index file; recall vs the exact brute-force answer")
            if not exact:
                print("bench: GPU graph walk DIFFERS from the oracle's", file=sys.stderr)
                rc = 4
    print(json.dumps(out), flush=True)
    if rc:
        sys.exit(rc)


class _DeviceBytes:
    """n bytes of device memory at `ptr` for torch.as_tensor (the CUDA array interface)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def gloo_exchange(torch, dist):
    """expann_exchange_fn for the rehearsal (ranks sharing one GPU): the rank's chunk goes to the
    host in stream order, gloo gathers, the gathered chunks go back on the same stream."""
    def fn(d_send, d_recv, nbytes, rank, world, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            mine = torch.as_tensor(_DeviceBytes(d_send, nbytes), device="cuda").cpu()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            torch.as_tensor(_DeviceBytes(d_recv, nbytes * world), device="cuda").copy_(torch.cat(parts))
            torch.cuda.current_stream().synchronize()
        return 0
    return fn


def gloo_alltoallv(torch, dist):
    """expann_alltoallv_fn for the rehearsal: the query slices of exchange pattern 2 over gloo (which has
    no all-to-all: one isend / irecv per peer), staged through the host in stream order."""
    def fn(d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes, rank, world, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            ops, landed = [], []
            for j in range(world):
                if send_bytes[j]:
                    part = torch.as_tensor(_DeviceBytes(d_send + send_off[j], send_bytes[j]), device="cuda").cpu()
                    ops.append(dist.P2POp(dist.isend, part, j))
                if recv_bytes[j]:
                    buf = torch.empty(recv_bytes[j], dtype=torch.uint8)
                    landed.append((j, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, j))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for j, buf in landed:
                torch.as_tensor(_DeviceBytes(d_recv + recv_off[j], recv_bytes[j]), device="cuda").copy_(buf)
            torch.cuda.current_stream().synchronize()
        return 0
    return fn


def fail(msg, code=5):
    print(f"bench: {msg}", file=sys.stderr, flush=True)
    sys.exit(code)


def main():
    a = parse()
    if a.workload == "c4":
        return bench_c4(a)
    import numpy as np
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = bool(os.environ.get("EXPANN_BENCH_REHEARSAL"))
    G = a.gpus
    if G < 1:
        fail("--gpus must be >= 1")
    # ---- how the G GPUs are driven: never fewer than asked for, never silently ----------------------
    #   launched by torch.distributed.run (WORLD_SIZE = G): one rank per GPU, the rank form of the C ABI
    #   launched plainly with --gpus G > 1: ONE process drives the G devices through the in-process
    #   handle (expann_sharded_create(devices 0..G-1), ncclCommInitAll: SURVEY 8e's form)
    if world > 1 and world != G:
        fail(f"WORLD_SIZE={world} but --gpus {G}: launch `python -m torch.distributed.run --nproc-per-node {G} "
             f"bench.py --gpus {G}` or plain `python bench.py --gpus {G}`")
    inproc = world == 1 and G > 1
    # (torch BEFORE the library: the library then binds to the HIP and RCCL copies torch bundles -- one runtime,
    # one RCCL in the process -- expann_amd/_lib.py; counting devices does not initialise the GPU)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # multi-process RCCL on this driver: dmabuf IPC only
    import torch
    n_dev = torch.cuda.device_count()
    need = G if inproc else (local_rank + 1 if world > 1 else 1)
    if n_dev < 1:
        fail(f"no HIP device visible; --gpus {G} needs {G} (libexpann_hip has no CPU fallback)")
    if n_dev < need and not rehearsal:
        fail(f"--gpus {G} but only {n_dev} HIP device(s) visible"
             + ("" if inproc else f" (this is local rank {local_rank})")
             + "; set EXPANN_BENCH_REHEARSAL=1 to let the shards share devices (a rehearsal, not a measurement)")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            # rehearsal of the rank grid on a box with fewer GPUs than ranks: gloo, ranks share GPUs
            local_rank = local_rank % torch.cuda.device_count()
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from expann_amd import GpuBruteForceEngine, ShardedBruteForceEngine

    # rank grid: pure row sharding by default (every rank searches every query); R < G = hybrid grid
    from expann_amd.sharded import shard_range, shard_grid, ceil_shard_range
    R, Q = shard_grid(G, a.row_shards or G)       # rank = query group x row shard
    if inproc and R != G:
        fail("the in-process form shards rows only (launch with torch.distributed.run for --row-shards)")
    # (rehearsal: ranks share a GPU, which RCCL refuses -- `--exchange native` then runs the same rank
    # form of the C ABI with the exchange handed in by the caller: expann_sharded_set_*_fn)
    native = world > 1 and R == G and a.exchange in (("native",) if rehearsal else ("auto", "native"))
    if a.exchange == "native" and world > 1 and not native:
        raise SystemExit("--exchange native needs pure row sharding")
    row_idx, qgroup = (0, 0) if inproc else (rank % R, rank // R)
    # rank r holds rows [r * ceil(N/R), min(N, (r+1) * ceil(N/R)))
    lo, hi = ceil_shard_range(a.n, row_idx, R)
    q_lo, q_hi = shard_range(a.m, qgroup, Q)
    m_local, pad = q_hi - q_lo, (a.m + Q - 1) // Q
    pattern_opt = {"auto": 0, "allgather": 1, "slices": 2}[a.exchange_pattern]

    # synthetic data, iid N(0,1), un-normalised (src/randomgeometry.h:87-95); fixed seeds
    gens = {}

    def gen_on(device):
        if device not in gens:
            gens[device] = torch.Generator(device=device)
        return gens[device]

    def make_rows(r_idx, device=dev):
        r_lo, r_hi = ceil_shard_range(a.n, r_idx, R)
        g = gen_on(device)
        g.manual_seed(1234 + r_idx)
        if a.dtype == "f32":
            b = torch.randn(r_hi - r_lo, a.d, device=device, dtype=torch.float32, generator=g)
            if a.sift_like:
                b = b.abs_().mul_(40).round_().clamp_(0, 255)
            return b
        if a.dtype == "i8":   # int8 uniform in [-127, 127]
            return torch.randint(-127, 128, (r_hi - r_lo, a.d), device=device, dtype=torch.int8, generator=g)
        # SURVEY 8d C4 stand-in: clamp(round(|N(0,1)|*40), 0, 255)
        return torch.randn(r_hi - r_lo, a.d, device=device, generator=g).abs_().mul_(40).round_() \
            .clamp_(0, 255).to(torch.uint8)

    base = make_rows(row_idx)
    g = gen_on(dev)
    g.manual_seed(4321)
    if a.dtype == "f32":
        queries = torch.randn(a.m, a.d, device=dev, dtype=torch.float32, generator=g)
        if a.sift_like:
            queries = queries.abs_().mul_(40).round_().clamp_(0, 255)
        if a.clustered and G == 1:
            # rows sorted by cluster: centres N(0, 1), members centre + 0.3 N(0, 1); queries near centres
            centres = torch.randn(a.clustered, a.d, device=dev, generator=g)
            per = (a.n + a.clustered - 1) // a.clustered
            base = base.mul_(0.3).add_(centres.repeat_interleave(per, 0)[:a.n])
            pick = torch.randint(0, a.clustered, (a.m,), device=dev, generator=g)
            queries = queries.mul_(0.3).add_(centres[pick])
    elif a.dtype == "i8":
        queries = torch.randint(-127, 128, (a.m, a.d), device=dev, dtype=torch.int8, generator=g)
    else:
        queries = torch.randn(a.m, a.d, device=dev, generator=g).abs_().mul_(40).round_() \
            .clamp_(0, 255).to(torch.float32)

    eng = None
    ip_state = None     # in-process form: per-shard tensors
    if inproc:
        # ONE process, G devices behind one handle; shard r's rows are generated on ITS device and adopted
        devices = [i % n_dev for i in range(G)]
        eng = ShardedBruteForceEngine(a.d, a.metric, a.dtype, devices=devices)
        bases, qs = [], []
        for r in range(G):
            d_r = torch.device("cuda", devices[r])
            r_lo, r_hi = ceil_shard_range(a.n, r, R)
            if r_hi == r_lo:
                break       # (the ceil partition left this and the later shards without rows)
            with torch.cuda.device(d_r):
                b_r = base if (r == 0 and d_r == dev) else make_rows(r, d_r)
                bases.append(b_r)
                eng.set_shard_device(r, b_r.data_ptr(), r_hi - r_lo, r_lo)
                qs.append(queries if d_r == dev else queries.to(d_r))
        if pattern_opt:
            eng.set_option("exchange_pattern", pattern_opt)
        G_act = eng.shards()
        outs = []
        for r in range(G_act):
            s_lo, s_hi = eng.slice(a.m, r)
            with torch.cuda.device(devices[r]):
                outs.append((torch.empty(max(1, s_hi - s_lo), a.k, dtype=torch.int64, device=qs[r].device),
                             torch.empty(max(1, s_hi - s_lo), a.k, dtype=torch.float32, device=qs[r].device)))
        for d_i in sorted(set(devices)):
            torch.cuda.synchronize(d_i)
        ip_state = (devices, bases, qs, outs, G_act)
    elif native:
        # one process per GPU, the exchange behind the C ABI: rank 0's RCCL unique id travels over
        # torch.distributed, ncclCommInitRank / the collectives / merge run inside libexpann_hip.
        # Every rank makes the SAME sequence of torch collectives whatever fails where: (1) rank 0's id --
        # or its failure -- is broadcast; (2) a flag is reduced BEFORE anyone enters ncclCommInitRank
        # (a rank that cannot take part would leave its peers blocked inside it); (3) one more after.
        pre_err = None
        box = [None]
        if rank == 0 and not rehearsal:
            try:
                box = [ShardedBruteForceEngine.unique_id()]
            except Exception as e:
                box = [("unique_id failed", str(e))]
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], tuple):
            pre_err = f"rank 0: {box[0][0]}: {box[0][1]}"
        try:    # this rank's own device is usable (what expann_sharded_create_rank does before the collective)
            probe = GpuBruteForceEngine(a.d, a.metric, a.dtype, device=local_rank)
            probe.close()
        except Exception as e:
            pre_err = pre_err or f"rank {rank}: {e}"
        bad = torch.tensor([1 if pre_err else 0], device="cpu" if rehearsal else dev, dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        err = pre_err
        if not int(bad.item()):
            try:
                eng = ShardedBruteForceEngine(a.d, a.metric, a.dtype, device=local_rank, rank=rank, world=G,
                                              unique_id=box[0])
                if rehearsal:
                    eng.set_exchange_fn(gloo_exchange(torch, dist))
                    eng.set_alltoallv_fn(gloo_alltoallv(torch, dist))
                if pattern_opt:
                    eng.set_option("exchange_pattern", pattern_opt)
                eng.set_shard_device(0, base.data_ptr() if hi > lo else 0, hi - lo, lo)
            except Exception as e:   # (reported, never silent: config.sharding names the exchange that ran)
                err = str(e)
            bad = torch.tensor([1 if err else 0], device="cpu" if rehearsal else dev, dtype=torch.int32)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if a.exchange == "native":
                raise SystemExit(f"--exchange native failed: {err or 'on another rank'}")
            print(f"bench: native RCCL exchange unavailable on some rank ({err or 'another rank'}); "
                  "using torch.distributed", file=sys.stderr)
            if eng is not None:
                eng.close()
            eng, native = None, False
    if eng is None:
        if hi == lo:
            fail(f"rank {rank} holds no rows (N = {a.n} over {R} row shards): only the native exchange "
                 "(the rank form of the C ABI) takes empty shards")
        eng = GpuBruteForceEngine(a.d, a.metric, a.dtype, device=local_rank)
        eng.set_base_device(base.data_ptr(), hi - lo, lo)
    for name, val in (("query_tile", a.query_tile), ("scan_kernel", a.scan_kernel), ("debug", a.debug),
                      ("sample_ratio", a.sample_ratio), ("sample_frac", a.sample_frac),
                      ("scan_chunks", a.scan_chunks), ("sample_run", a.sample_run)):
        if val:
            eng.set_option(name, val)
    if a.xcd_tolerance >= 0:
        eng.set_option("xcd_tolerance", a.xcd_tolerance)
    from expann_amd.sharded import GridShardedSearch, chunk_bytes, unpack_chunk
    from expann_amd import merge_topk_strided_device
    bufs = {}
    # everything of a step -- the engine's kernels, the merge, the collectives' stream dependencies --
    # is ordered on ONE explicit stream (torch's default stream has handle 0, which the engine would
    # replace by a stream of its own)
    torch.cuda.synchronize()
    work_stream = torch.cuda.Stream(device=dev)
    stream = work_stream.cuda_stream
    use_async = not a.sync_search
    # deferred check: a search is enqueued without a host wait, so the next kernels / the RCCL
    # exchange are already queued when it ends; expann_sync validates all steps at the end
    eng.set_option("async_search", 1 if use_async else 0)
    cb = chunk_bytes(pad, a.k)

    def alloc(name, nbytes, like):
        if name not in bufs:
            bufs[name] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if name == "mine":      # rows past this rank's query slice stay padding for good
                pi, pd = unpack_chunk(bufs[name], pad, a.k)
                pi.fill_(-1)
                pd.fill_(float("inf"))
        return bufs[name]

    def local_search(q, k, chunk):
        base_ptr = chunk.data_ptr()
        eng.search_device(q.data_ptr(), q.shape[0], k, base_ptr, base_ptr + pad * k * 8, stream)

    def merge(gathered, n_lists, rows, k, out_chunk):
        gp, op = gathered.data_ptr(), out_chunk.data_ptr()
        merge_topk_strided_device(local_rank, gp, gp + rows * k * 8, cb // 8, cb // 4, n_lists, rows, k,
                                  op, op + rows * k * 8, stream)

    def sync_devices():
        if inproc:
            for d_i in sorted(set(ip_state[0])):
                torch.cuda.synchronize(d_i)
        else:
            torch.cuda.synchronize()

    if inproc:
        devices, bases, qs, outs, G_act = ip_state
        q_ptrs = [q.data_ptr() for q in qs]
        i_ptrs = [o[0].data_ptr() for o in outs]
        d_ptrs = [o[1].data_ptr() for o in outs]

        def step():
            # every shard's scan, the exchange and the merge of its query slice: enqueued on all G devices
            # (one host thread per device inside the library), nothing waits
            eng.search_devices(q_ptrs, a.m, a.k, i_ptrs, d_ptrs)
            return None, None

        def collect():
            parts = [eng.slice(a.m, r) for r in range(G_act)]
            ids = torch.cat([outs[r][0][:b - c].to(dev) for r, (c, b) in enumerate(parts)], 0)
            dd = torch.cat([outs[r][1][:b - c].to(dev) for r, (c, b) in enumerate(parts)], 0)
            return ids, dd
    elif native:
        out_ids = torch.empty(a.m, a.k, dtype=torch.int64, device=dev)
        out_d = torch.empty(a.m, a.k, dtype=torch.float32, device=dev)

        def step():
            # local scan, exchange, merge: all enqueued on `stream` by the library
            eng.search_device(queries.data_ptr(), a.m, a.k, out_ids.data_ptr(), out_d.data_ptr(), stream)
            return out_ids, out_d
    elif world > 1 and R == G and pattern_opt == 2:
        # --exchange torch --exchange-pattern slices: pattern 2 through torch.distributed
        from expann_amd.sharded import SliceShardedSearch

        def merge_slices(l_ids, l_d, n_lists, per, cnt, k, o_ids, o_d):
            merge_topk_strided_device(local_rank, l_ids.data_ptr(), l_d.data_ptr(), per * k, per * k, n_lists, cnt, k,
                                      o_ids.data_ptr(), o_d.data_ptr(), stream)

        def local_all(q, k, chunk):
            p0 = chunk.data_ptr()
            eng.search_device(q.data_ptr(), q.shape[0], k, p0, p0 + q.shape[0] * k * 8, stream)

        def alloc_plain(name, nbytes, like):
            if name not in bufs:
                bufs[name] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            return bufs[name]

        ss = SliceShardedSearch(dist, world, rank, local_all, merge_slices, alloc_plain)

        def step():
            with torch.cuda.stream(work_stream):
                return ss.search(queries, a.k)
    else:
        ss = GridShardedSearch(dist if world > 1 else None, world, rank, R, local_search, merge, alloc)

        def step():
            with torch.cuda.stream(work_stream):
                return ss.search(queries, a.k)

    def timed(n_steps):
        """(seconds, ok): ok is False when a deferred search of the loop needed the synchronous
        retry (agreed between the ranks: every rank then repeats the steps the waiting way)"""
        if world > 1:
            dist.barrier()
        sync_devices()
        t_start = time.perf_counter()
        for _ in range(n_steps):
            step()
        ok = True
        if use_async:
            try:
                eng.sync()      # every deferred search of the loop is complete and valid
            except RuntimeError as e:
                print(f"bench: {e}", file=sys.stderr)
                ok = False
        sync_devices()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t_start
        if world > 1:
            flag = torch.tensor([0 if ok else 1], device="cpu" if rehearsal else dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            ok = int(flag.item()) == 0
        return dt, ok

    # one-time lazy initialisation outside any count: derived copies of the index (fp16 / uint8),
    # workspace allocation and -- with G > 1 -- the communicators
    _, ok = timed(1)
    if ok:
        _, ok = timed(a.warmup)
    eng.set_profiling(True)
    if ok:
        elapsed, ok = timed(a.steps)
    if not ok:   # (never on the iid workloads) a deferred search needed the retry: time the waiting form
        print("bench: repeating the timed steps with --sync-search", file=sys.stderr)
        use_async = False
        eng.set_option("async_search", 0)
        eng.get_profile()
        timed(1)
        elapsed, _ = timed(a.steps)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.get_profile()
    enqueue_ms = eng.last_enqueue_ms() if inproc else None
    eng.set_profiling(False)
    # one more step, outside the timed region, whose result the checks below look at (it is the
    # same search on the same inputs as every timed step)
    got_ids, got_d = step()
    if use_async:
        eng.sync()
    sync_devices()
    if inproc:
        got_ids, got_d = collect()
    got_ids, got_d = got_ids[:a.m], got_d[:a.m]

    def whole_base():
        if R == 1:
            return base
        return torch.cat([base if r == row_idx else make_rows(r) for r in range(R)
                          if ceil_shard_range(a.n, r, R)[1] > ceil_shard_range(a.n, r, R)[0]], 0)

    if a.verify and G > 1:
        if rank == 0:
            whole = whole_base()
            ref = GpuBruteForceEngine(a.d, a.metric, a.dtype, device=local_rank)
            ref.set_base_device(whole.data_ptr(), a.n, 0)
            ref_ids = torch.empty(a.m, a.k, dtype=torch.int64, device=dev)
            ref_d = torch.empty(a.m, a.k, dtype=torch.float32, device=dev)
            ref.search_device(queries.data_ptr(), a.m, a.k, ref_ids.data_ptr(), ref_d.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(got_ids, ref_ids)) and \
                bool(torch.equal(got_d.view(torch.int32), ref_d.view(torch.int32)))
            print(f"verify: sharded rows/{R} x queries/{Q} result "
                  f"{'IDENTICAL to' if same else 'DIFFERS from'} the unsharded search", file=sys.stderr)
            ref.close()
            del whole
            if not same:
                sys.exit(3)

    rc = 0
    if rank == 0:
        ms_per_step = elapsed * 1e3 / a.steps
        qps = a.m * a.steps / elapsed
        launches = max(1, prof["scan_launches"])
        scan_ms = prof["scan_ms"] / launches
        n_local = hi - lo
        passes = prof["scan_query_tiles"] / launches
        esz = 4 if a.dtype == "f32" else 1
        alg_bytes = passes * n_local * a.d * esz        # one pass of a query tile = N*d*sizeof B
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        default_cfg = (a.n, a.d, a.m, a.k, G) == (1_000_000, 128, 10_000, 10, 1)
        traffic, traffic_src = profiled_traffic(prof["scan_kernel"]) if default_cfg else (None, None)
        # SURVEY 8d's pass accounting (one pass of a query tile over the base = N*d*sizeof bytes).  With
        # 256-query tiles the passes are served from L2 / LDS, not HBM: the passes-times-bytes rate is
        # NOT an HBM figure and is only priced against the HBM peak when the kernel really streams
        # every pass from HBM (a handful of queries: bound "hbm" below); `traffic` is the measured HBM side.
        hbm_view = {"passes_per_launch": passes, "query_tile": int(prof["query_tile"]),
                    "bytes_per_pass": n_local * a.d * esz,
                    "single_pass_equiv_GBps": round(n_local * a.d * esz / (scan_ms * 1e-3) / 1e9, 2)
                    if scan_ms > 0 else 0.0}
        if prof["scan_kernel"].startswith("scan_gemm_i8"):
            ops = 2.0 * n_local * a.d * m_local             # C5 2*N*d*m int ops
            tops = ops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS,
                        "unit": "TOP/s", "frac": round(tops / MFMA_I8_PEAK_TOPS, 4)}
        elif prof["scan_kernel"].startswith("scan_direct_f16"):
            # a handful of queries: the fp16 filter streamed from HBM (scan_direct_f16.hpp) reads the
            # scaled fp16 copy (2 B per element) + one 4-byte row term per row and pass
            gbs = passes * n_local * (a.d * 2 + 4) / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
            roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "note": "bytes = passes x N x (2 d + 4): the fp16 rows the filter reads; "
                                "pass_accounting counts the fp32 rows of SURVEY 8d (N*d*4)"}
        elif prof["scan_kernel"].startswith("scan_gemm_f16"):
            # one fp16 MFMA product per fp32 product (scaled operands, rigorous slack, exact
            # re-rank): executed flops = algorithmic 2*N*d*m, priced against the dense fp16 peak
            alg = 2.0 * n_local * a.d * m_local
            tf = alg / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                        "mfma_dtype": "fp16 (filter only; results exact after fp32 re-rank)",
                        "algorithmic_vs_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 3)}
            if passes <= 2:
                # a handful of queries: one or two passes over the fp16 rows, the HBM read is the bound
                gbs = passes * n_local * a.d * 2 / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
                roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "note": "small batch: bytes = query tiles x N x d x 2 (the fp16 rows the filter reads)"}
        elif prof["scan_kernel"].startswith("scan_gemm_bf16x3"):
            # fp32 products evaluated exactly enough on the bf16 cores as 3 bf16 MFMA products
            # (hi*hi + hi*lo + lo*hi): executed flops = 3 x the algorithmic 2*N*d*m
            alg = 2.0 * n_local * a.d * m_local
            tf_alg = alg / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(3 * tf_alg, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(3 * tf_alg / MFMA_BF16_PEAK_TFLOPS, 4),
                        "mfma_dtype": "bf16 (3-term split of fp32 operands, results exact after re-rank)",
                        "mfma_products_per_fp32_product": 3,
                        "algorithmic_TFLOPs": round(tf_alg, 1),
                        "algorithmic_vs_fp32_mfma_peak": round(tf_alg / MFMA_F32_PEAK_TFLOPS, 3)}
        elif prof["scan_kernel"].startswith("scan_gemm"):
            # GEMM-form filter on the matrix cores: algorithmic flops = 2*N*d*m
            flops = 2.0 * n_local * a.d * m_local
            tf = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
        else:
            roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)}
        roofline.update({
            "traffic": round(traffic / 1e9, 2) if traffic else None,
            "traffic_unit": "GB per launch (rocprofv3 PMC, FETCH_SIZE x2 calibrated + WRITE_SIZE)",
            "traffic_source": traffic_src, "kernel": prof["scan_kernel"],
            "kernel_ms": round(scan_ms, 4), "launches": int(launches),
            "algorithmic_bytes_per_launch": alg_bytes, "pass_accounting": hbm_view,
            "candidates_per_query": round(prof["candidates"] / m_local, 1)})
        desc = {"f32": "fp32", "i8": "int8", "u8": "uint8"}[a.dtype]
        shape = f"{a.n // 1_000_000}M" if a.n % 1_000_000 == 0 else str(a.n)
        deferred = int(prof.get("deferred_searches", 0))
        if use_async and deferred >= launches:
            host_sync = ("deferred: searches are enqueued back to back, one expann_sync validates all K "
                         "steps inside the timed region")
        elif use_async and deferred:
            host_sync = f"mixed: {deferred} of {int(launches)} timed searches deferred their check"
        else:
            host_sync = "per step" + (" (this path checks its flags at once: exact-uint8 shortcut / retry)"
                                      if use_async else "")
        sharded_handle = inproc or native
        transport = {0: "none", 1: "RCCL", 2: "device copies (shards share a device)",
                     3: "the caller's transport (gloo)"}[eng.exchange()] if sharded_handle else "torch.distributed"
        pattern = {0: "none", 1: "one all-gather of whole [m][k] chunks, every rank merges all queries",
                   2: "all-to-all of query slices, each rank merges m/G queries"
                      + ("" if inproc else ", all-gather of the merged slices")}[eng.exchange_pattern()] \
            if sharded_handle else "all_gather + expann_merge_topk"
        launch = ("one process, one handle over all devices (expann_sharded_create, ncclCommInitAll; one enqueue "
                  "thread per device)" if inproc else
                  ("one rank per GPU (torch.distributed.run), the rank form of the C ABI "
                   "(expann_sharded_create_rank, ncclCommInitRank)" if native else
                   ("one rank per GPU (torch.distributed.run), exchange through torch.distributed" if world > 1
                    else "one process, one GPU")))
        rows_per_rank = [ceil_shard_range(a.n, r, R)[1] - ceil_shard_range(a.n, r, R)[0] for r in range(R)]
        out = {"metric": f"queries/sec at recall@{a.k}=1.0 (exact brute force), {shape}xd{a.d} {desc}, k={a.k}",
               "value": round(qps, 1), "unit": "queries/s", "n_gpus": G, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"brute_force_engine {a.metric.upper()} N={a.n} d={a.d} {desc}, "
                                      f"{a.m} batched queries, k={a.k} (BASELINE "
                                      f"{ {'c2': 'configs[1]', 'c3': 'configs[2]', 'c5': 'configs[4]'}[a.workload]})",
                          "n": a.n, "d": a.d, "m": a.m, "k": a.k, "metric": a.metric,
                          "host_sync": host_sync,
                          "launch": launch,
                          "devices_visible": n_dev,
                          # the communicator's size as RCCL itself reports it (ncclCommCount); 0 = no RCCL communicator
                          "comm_ranks": (eng.comm_ranks() if sharded_handle else (world if world > 1 and not rehearsal else 0)),
                          "rows_per_rank": rows_per_rank if G > 1 else [a.n],
                          "sharding": (f"rows/{R} x queries/{Q}: rank r scans rows [r*ceil(N/{R}), ...) for "
                                       f"{'all' if Q == 1 else 'its slice of the'} queries; transport = {transport}; "
                                       f"exchange = {pattern}; rank 0 scans {n_local} rows for {m_local} queries, "
                                       f"roofline figures are rank 0's launch") if G > 1 else "none"},
               "roofline": roofline,
               "verified_step": "one extra step after the timed loop (the same search on the same inputs)"}
        if rehearsal:
            out["config"]["rehearsal"] = ("EXPANN_BENCH_REHEARSAL: shards / ranks share the visible GPU(s); "
                                          "a functional rehearsal, not a multi-GPU measurement")
        if enqueue_ms is not None:
            out["config"]["host_enqueue_ms_per_step"] = round(enqueue_ms, 4)
        # ---- the bench proves its own claim: GPU result vs the CPU oracle -----------------------
        checked = None
        mname = {("f32", "l2"): "METRIC_L2_F32", ("f32", "ip"): "METRIC_IP_F32",
                 ("i8", "l2"): "METRIC_L2_I8", ("i8", "ip"): "METRIC_IP_I8",
                 ("u8", "l2"): "METRIC_L2_U8"}[(a.dtype, a.metric)]
        base_host = queries_host = None
        if G == 1 and not a.no_cpu_baseline:
            try:
                base_host, queries_host = base.cpu().numpy(), queries.cpu().numpy()
                out["cpu_baseline"], checked = cpu_baseline(base_host, queries_host, a.k, a.cpu_seconds, mname)
            except Exception as e:  # the baseline is a reported extra, never the product
                out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0,
                                       "kind": "port", "sample": f"failed: {e}"}
        if checked is None and not a.no_verify:
            nv = a.verify_queries or 8
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle_ctypes as oc
                if base_host is None:
                    if inproc:
                        base_host = np.concatenate([b.cpu().numpy() for b in ip_state[1]], 0)
                    else:
                        base_host = whole_base().cpu().numpy()
                    queries_host = queries.cpu().numpy()
                checked = oc.brute_force(base_host, queries_host[:nv], a.k, getattr(oc, mname),
                                         min(nv, os.cpu_count() or 1))
            except Exception as e:
                out["verify_error"] = str(e)
        if checked is not None and not a.no_verify:
            recall, nq, exact = oracle_check(checked[0], checked[1], got_ids.cpu().numpy(), got_d.cpu().numpy())
            out["recall_measured"] = round(recall, 6)
            out["verified_queries"] = nq
            out["bit_exact"] = exact
            out["verified_against"] = ("oracle/expann_oracle.c (CPU restatement of src/brute_force_engine.h:28-46 "
                                       "+ src/distance.h), same inputs, ids and distance bits")
            if not exact or recall != 1.0:
                print(f"bench: GPU result DIFFERS from the oracle on {nq} checked queries "
                      f"(recall {recall:.6f}, bit_exact {exact})", file=sys.stderr)
                rc = 4
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
