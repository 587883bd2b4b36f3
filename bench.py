#!/usr/bin/env python
"""bench.py — CelebBasis personalization training step on MI355X (BASELINE.json metric, config 2).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...            # re-executes itself under torch.distributed.run, one rank per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full personalization training step of ONE identity at SD v1-4 / 512x512 / bs=1 / fp32:
VAE encode -> FR-net + MLP + basis (+EMA) -> CLIP text (with the manager hook) -> q_sample -> UNet fwd -> MSE ->
input-gradient backward through UNet and CLIP -> dW/db -> AdamW. Nothing is cached or skipped inside the timed region.
Inputs (a replay stream of synthetic batches) are resident in HBM before the timed region starts.
With N > 1 every rank personalises its own identity (SURVEY.md §8 e: independent units, no data-path collective); the only
collective is the checkpoint-merge all_gather after the timed region (RCCL), whose result is checked (`merged_identities == N`).
`value` = steps of all ranks / max-over-ranks time; `identities_per_sec` = value / 800 (aigc_id.yaml:135 max_steps).

Prints ONE JSON line on rank 0 (contract in the task statement) with
  `roofline`      dominant kernel of the fp32 step, measured live with HIP events around every GEMM-class launch of one extra
                  step. Numbers of the committed rocprofv3 trace / PMC passes (profiles/r06*) are added next to the live ones ONLY
                  when their stamp (sha of the kernel sources + plan table, `build`) is this tree's; otherwise `profile_stale` is
                  true and the line carries live measurements alone;
  `cpu_baseline`  the CPU oracle (a from-scratch restatement = kind "port") timed on this box's host cores on a bounded sample;
  `config3`       secondary measurement, N == 1 only: BASELINE config 3 (bf16 UNet / VAE + fp32 coefficient gradient, bs=4) with its
                  own dominant-kernel roofline against the 2.5 PF dense bf16 MFMA peak. Never used for `value`.

`--dry` runs the same control flow on dry contexts with the gloo backend (no GPU: launcher / sharding / merge / JSON contract
only, computes nothing) — used by the CPU test of the N > 1 launcher path.
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16, dense (not the 2:1-sparsity headline)
PEAK_HBM_GBPS = 8000.0
STEPS_PER_IDENTITY = 800           # aigc_id.yaml:135 max_steps


def trace_symbols(kernel):
    """profiler label (`gemm_glds_kernel<float,128,128,CONVF>`) -> the rocprofv3 symbols it may run as, as tools/prof_stats.py / pmc_traffic.py
    write them. The LDS-DMA kernels exist in a scalar-offset (FAST) and a per-lane form: one label, up to two symbols."""
    import re
    kinds = {"ROW": 0, "COL": 1, "CONV": 2, "CONVF": 3}
    m = re.match(r"gemm_kernel<(\d+),(\d+),(\w+),(\w+)(,generic)?>", kernel)
    m4 = re.match(r"gemm_kernel_occ4<(\w+),(\w+)>", kernel)
    mg = re.match(r"gemm_glds_kernel<(float|bf16),(\d+),(\d+),(\w+)>", kernel)       # <T, BM, BN, AK, row bytes, stages, FAST>
    m80 = re.match(r"gemm_glds80_kernel<float,(\d+),80,(\w+)>", kernel)               # <BM, AK, stages, FAST>
    mp8 = re.match(r"gemm_bf16_p8_kernel<bf16,(\d+),(\d+),(\w+)>", kernel)           # <AK, waves along M, ablation>
    ak3 = {"ROW": 0, "CONV": 1, "CONVF": 2}
    if mp8:
        return [f"gemm_bf16_p8_kernel<{ak3[mp8.group(3)]}, {2 if mp8.group(2) == '256' else 4}, 0>"]
    if mg:
        t = "float" if mg.group(1) == "float" else "__bf16"
        return [f"gemm_glds_kernel<{t}, {mg.group(2)}, {mg.group(3)}, {ak3[mg.group(4)]}, 128, 2, {f}>" for f in ("true", "false")]
    if m80:
        return [f"gemm_glds80_kernel<{m80.group(1)}, {ak3[m80.group(2)]}, 2, {f}>" for f in ("true", "false")]
    if m:
        return [f"gemm_kernel<{m.group(1)}, {m.group(2)}, {kinds[m.group(3)]}, {kinds[m.group(4)]}, {'false' if m.group(5) else 'true'}>"]
    if m4:
        return [f"gemm_kernel_occ4<{kinds[m4.group(1)]}, {kinds[m4.group(2)]}>"]
    return []


def tree_stamp():
    from celebbasis_amd.build import source_digest
    return source_digest()


def pmc_traffic(kernel, bf16=False):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need
    separate passes and cannot be collected inside this process): profiles/r06*_pmc_hbm_traffic{,_config3}.json, written by
    tools/pmc_traffic.py from `rocprofv3 --kernel-trace --pmc ... -- python bench.py`. Corrected as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950: FETCH_SIZE tallies the 128-B requests of 16-B/lane reads and `buffer_load ... lds` at 64 B, so
    traffic = 2 x FETCH_SIZE + WRITE_SIZE. The file is used only when it was measured on THIS tree (stamp = sha of the kernel
    sources + plan table); otherwise (None, {"stale": ...}). Returns (bytes | None, detail)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r06*_pmc_hbm_traffic{'_config3' if bf16 else ''}.json")))
    syms = trace_symbols(kernel)
    if not files or not syms:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("stamp") != tree_stamp():
        return None, dict(source=os.path.basename(files[-1]), stale=True, file_stamp=d.get("stamp"), tree_stamp=tree_stamp())
    cands = [(s_, d.get("per_symbol", {}).get(s_)) for s_ in syms]
    cands = [(s_, e) for s_, e in cands if e and e["fetch_bytes_per_launch"] is not None and e["write_bytes_per_launch"] is not None]
    if not cands:
        return None, None
    sym, e = max(cands, key=lambda c: c[1]["launches"])            # the form that carries the launches (FAST on every SD v1-4 shape)
    corrected = 2 * e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    detail = dict(source=os.path.basename(files[-1]), symbol=sym, launches=e["launches"], stale=False,
                  fetch_size_bytes_per_launch_raw=round(e["fetch_bytes_per_launch"]), write_size_bytes_per_launch=round(e["write_bytes_per_launch"]),
                  note="traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts the 128-B requests of 16-B/lane and LDS-DMA reads "
                       "at 64 B); L2-miss traffic that the 256 MiB Infinity Cache may still serve, i.e. an upper bound of HBM bytes proper")
    return round(corrected), detail


# kernel symbols behind each HBM-bound profiler group (rocprofv3 names, `cb::` stripped as tools/prof_stats.py writes them)
HBM_GROUP_SYMBOLS = {
    "groupnorm_fwd": ("gn_partial_kernel", "gn_final_kernel", "gn_apply_kernel", "gn_fused_fwd_kernel"),
    "groupnorm_bwd": ("gn_bwd_partial_kernel", "gn_bwd_final_kernel", "gn_bwd_apply_kernel", "gn_fused_bwd_kernel"),
    "layernorm_fwd": ("layernorm_fwd_kernel",), "layernorm_bwd": ("layernorm_bwd_kernel",),
    "splitk_reduce_kernel": ("splitk_reduce_kernel",),
}


def rocprof_trace(bf16=False, prefix="r06"):
    """the newest committed rocprofv3 kernel trace of this leg — profiles/<prefix>*_bench_kernel_stats.csv = `bench.py --no-config3`
    (the fp32 step alone), profiles/<prefix>*_config3_kernel_stats.csv = `bench.py --trace-leg config3` — if and only if its
    stamp line says it was taken on THIS tree. Returns (file | None, stale, {symbol: (calls_per_step, ms_per_step, avg_us)},
    {base name: ms_per_step})."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{prefix}*_{'config3' if bf16 else 'bench'}_kernel_stats.csv")))
    if not files:
        return None, None, {}, {}
    with open(files[-1]) as f:
        lines = f.readlines()
    st = tree_stamp()
    head = next((l for l in lines if l.startswith("# stamp")), "")
    kv = dict(t.split("=", 1) for t in head.split()[2:] if "=" in t)
    if kv.get("src_sha") != st["src_sha"] or kv.get("plan_table_sha") != st["plan_table_sha"]:
        return os.path.basename(files[-1]), True, {}, {}
    nsteps = int(kv.get("nsteps", "9"))
    per, base = {}, {}
    for r in csv.DictReader([l for l in lines if not l.startswith("#")]):
        per[r["name"]] = (int(r["calls"]) / nsteps, float(r["ms_per_step"]), float(r["avg_us"]))
        b_ = r["name"].split("<")[0]
        base[b_] = base.get(b_, 0.0) + float(r["ms_per_step"])
    return os.path.basename(files[-1]), False, per, base


def maybe_self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become `torch.distributed.run` with N ranks"""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvp(cmd[0], cmd)


def timed_loop(run, steps, warmup, world, dist, dev, dry):
    for i in range(warmup):
        run(i)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        run(warmup + i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    PER_RANK.clear()
    if world > 1:
        # every rank's own clock next to the MAX the contract asks for (a slow GPU / a late rank shows up in the line itself), plus its HBM use
        free, total = (0, 0) if dry else torch.cuda.mem_get_info()
        mine = torch.tensor([elapsed, float(total - free) / 2 ** 30], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        PER_RANK.extend(dict(rank=r, elapsed_s=round(float(t[0]), 4), hbm_used_gib=round(float(t[1]), 2)) for r, t in enumerate(every))
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


PER_RANK = []


def profile_step(ctx, run, peak_tflops, bf16=False):
    """one extra step with a HIP-event pair around every GEMM-class launch -> (roofline dict, phases, stats).
    Live events are the evidence for `achieved` (the dominant kernel's launches last 100+ us). An event pair around a sub-30 us
    launch reads long (fixed ~5 us of dispatch + signal latency per pair), so every per-variant row ALSO carries `ms_trace` — the
    rocprofv3 per-symbol time of the committed trace — but only when that trace was taken on this very tree (`profile_stale`
    false); a stale trace is never mixed into the line."""
    ctx.reset_stats()
    ctx.profile_begin()
    run(0)
    prof = ctx.profile_end()
    st = ctx.stats()
    phases = {p["kernel"][6:]: round(p["ms"], 3) for p in prof if p["kernel"].startswith("phase:")}
    prof = [p for p in prof if not p["kernel"].startswith("phase:")]
    hbm = [p for p in prof if p.get("bytes", 0) > 0 and p["flops"] == 0]    # HBM-bound groups: algorithmic bytes
    prof = [p for p in prof if p["flops"] > 0]                             # MFMA-bound launches (GEMM family + fused attention)
    tot_ms = sum(p["ms"] for p in prof); tot_fl = sum(p["flops"] for p in prof)
    # dominant kernel = the variant that carries the most algorithmic FLOPs of the step
    dom = max(prof, key=lambda p: p["flops"])
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    src, stale, per_sym, sym_ms = rocprof_trace(bf16)

    # fused-attention profiler rows span several kernel symbols (weights: the share of a symbol that belongs to the row — the v2 transposes
    # run once per forward and three times per backward launch)
    n_f2f = sum(q["launches"] for q in prof if q["kernel"] == "flash_bf16_v2_fwd_kernel")
    n_f2b = sum(q["launches"] for q in prof if q["kernel"] == "flash_bf16_v2_bwd_kernels")
    tshare = n_f2f / max(n_f2f + 3 * n_f2b, 1)
    FLASH_SYMS = {
        "flash_fwd_kernel": [("flash_fwd_kernel", 1.0), ("flash_fwd_combine_kernel", 1.0)],
        "flash_bwd_kernels": [("flash_bwd_dq_kernel", 1.0), ("flash_bwd_dkv_kernel", 1.0), ("flash_dkv_reduce_kernel", 1.0), ("flash_dq_combine_kernel", 1.0),
                              ("flash_dvec_kernel", 1.0)],
        "flash_bf16_fwd_kernel": [("flash_bf16_fwd_kernel", 1.0)],
        "flash_bf16_bwd_kernels": [("flash_bf16_dq_kernel", 1.0), ("flash_bf16_dkv_kernel", 1.0), ("flash_bf16_dkv_reduce_kernel", 1.0), ("flash_bf16_dvec_kernel", 1.0)],
        "flash_bf16_v2_fwd_kernel": [("f2_fwd_kernel", 1.0), ("f2_transpose_kernel", tshare)],
        "flash_bf16_v2_bwd_kernels": [("f2_dq_kernel", 1.0), ("f2_dkv_kernel", 1.0), ("f2_transpose_kernel", 1.0 - tshare)],
    }

    def trace_ms(p):
        """(ms per step, launches per step) of a profiler row in the matching trace, or (None, None)"""
        hit = [per_sym[s_] for s_ in trace_symbols(p["kernel"]) if s_ in per_sym]
        if hit:
            return sum(h[1] for h in hit), sum(h[0] for h in hit)
        grp = FLASH_SYMS.get(p["kernel"])
        if grp and any(b_ in sym_ms for b_, _ in grp):
            return sum(sym_ms.get(b_, 0.0) * w for b_, w in grp), None
        return None, None

    hb = []
    for p in sorted(hbm, key=lambda p: -p["ms"]):
        hb.append(dict(kernel=p["kernel"], launches=p["launches"], algorithmic_mb=round(p["bytes"] / 1e6, 1), ms_events=round(p["ms"], 3)))
    # rocprofv3 view of the HBM-bound groups: the trace has one duration per symbol, so a group's size classes and its SiLU / plain
    # variants (same GroupNorm symbols) are merged
    groups = {}
    for p in hbm:
        g = p["kernel"].split("[")[0].replace("<bf16>", "").replace("groupnorm_silu_", "groupnorm_")
        a = groups.setdefault(g, dict(bytes=0.0, launches=0))
        a["bytes"] += p["bytes"]; a["launches"] += p["launches"]
    rp = []
    for g, a in groups.items():
        syms = [s for s in HBM_GROUP_SYMBOLS.get(g, ()) if s in sym_ms]
        if not syms:
            continue
        ms = sum(sym_ms[s] for s in syms)
        rp.append(dict(kernel=g, symbols=list(syms), launches=a["launches"], algorithmic_mb=round(a["bytes"] / 1e6, 1), ms_rocprof=round(ms, 3),
                       achieved_gbps=round(a["bytes"] / (ms * 1e-3) / 1e9, 1), frac_of_8tbps=round(a["bytes"] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)))
    dom_tr, dom_tr_launches = trace_ms(dom)
    rows = []
    for p in sorted(prof, key=lambda p: -p["flops"]):
        tms, tl = trace_ms(p)
        rows.append(dict(kernel=p["kernel"], launches=p["launches"], gflop=round(p["flops"] / 1e9, 1), ms_events=round(p["ms"], 3),
                         ms_trace=None if tms is None else round(tms, 3), launches_trace=None if tl is None else round(tl, 2),
                         tflops_trace=None if not tms else round(p["flops"] / (tms * 1e-3) / 1e12, 2),
                         # algorithmic operand + result bytes (A once, B once, C once) and the rate they move at: for the M <= 256 problems
                         # (the 16-row 80-column tiles) that is the weight stream, and neither roofline is near (round-5 verdict item 5)
                         algorithmic_mb=round(p.get("bytes", 0) / 1e6, 1),
                         gbps_trace=None if not tms or not p.get("bytes") else round(p["bytes"] / (tms * 1e-3) / 1e9, 1)))
    covered = [r for r in rows if r["ms_trace"]]
    roofline = dict(bound="mfma", kernel=dom["kernel"], achieved=round(ach, 2), peak=peak_tflops, unit="TFLOP/s",
                    frac=round(ach / peak_tflops, 4), traffic=None, traffic_detail=None,
                    launches_per_step=dom["launches"], avg_launch_ms=round(dom["ms"] / dom["launches"], 5),
                    algorithmic_gflop_per_step=round(dom["flops"] / 1e9, 1),
                    algorithmic_bytes_per_launch=round(dom.get("bytes", 0) / dom["launches"]),
                    profile_source=src, profile_stale=stale,
                    rocprof=None if dom_tr is None else dict(
                        launches_per_step=dom_tr_launches, avg_launch_ms=round(dom_tr / max(dom_tr_launches or dom["launches"], 1), 5),
                        achieved=round(dom["flops"] / (dom_tr * 1e-3) / 1e12, 2), frac=round(dom["flops"] / (dom_tr * 1e-3) / 1e12 / peak_tflops, 4),
                        note="same algorithmic FLOPs over the committed rocprofv3 trace's total time for this symbol (trace taken on this tree)"),
                    all_gemm_kernels=dict(gflop_per_step=round(tot_fl / 1e9, 1), ms_events=round(tot_ms, 3),
                                          achieved_events=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                          ms_trace=round(sum(r["ms_trace"] for r in covered), 3) if covered else None,
                                          achieved_trace=round(sum(r["gflop"] for r in covered) / sum(r["ms_trace"] for r in covered), 2) if covered else None,
                                          trace_covers_gflop=round(sum(r["gflop"] for r in covered), 1) if covered else None,
                                          note="event pairs around sub-30 us launches read long: *_events is a lower bound of the rate; *_trace "
                                               "(present only with a trace of this tree) is the per-symbol rocprofv3 time"),
                    attention_mfma=[dict(kernel=r["kernel"], launches=r["launches"], gflop=r["gflop"], ms_events=r["ms_events"], ms_trace=r["ms_trace"],
                                         frac_of_peak_events=round(r["gflop"] / r["ms_events"] / peak_tflops, 4),
                                         frac_of_peak_trace=None if not r["ms_trace"] else round(r["gflop"] / r["ms_trace"] / peak_tflops, 4))
                                    for r in rows if r["kernel"].startswith("flash")],
                    by_kernel=rows,
                    hbm_bound=dict(note="bandwidth of the HBM-bound groups is quoted from the rocprofv3 per-symbol times of a trace of THIS tree only "
                                        "(`rocprof`, empty when profile_stale); `events` lists the live event-pair sums, which read ~2.5x long on 5 us kernels",
                                   rocprof=rp, events=hb))
    return roofline, phases, st


def frac_range(value, bf16=False):
    """roofline.frac of the dominant kernel over the committed bench lines of THIS tree (profiles/BENCH_r06_run*.json with the same
    stamp: other boxes of the pool) plus this run: [min, max, n]"""
    import glob
    st, vals = tree_stamp(), [value]
    for p in glob.glob(os.path.join(ROOT, "profiles", "BENCH_r06_run*.json")):
        try:
            d = json.load(open(p))
            if d.get("build") == st:
                r = d["config3"]["roofline"] if bf16 else d["roofline"]
                vals.append(r["frac"])
        except Exception:
            pass
    return [min(vals), max(vals), len(vals)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trace-leg", choices=["all", "config3"], default="all",
                    help="diagnostics for rocprofv3 traces (tools/gpurun.sh trace): 'config3' runs ONLY the config-3 leg, so that the trace's "
                         "per-symbol times are not mixed with the fp32 leg's (the JSON line then carries no config-2 value)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary, cross-step-pipelined measurement")
    ap.add_argument("--no-config3", action="store_true", help="skip the secondary bf16 bs=4 measurement (BASELINE config 3)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=64, help="host threads for the CPU baseline (reported as `cores`)")
    ap.add_argument("--batches", type=int, default=4, help="distinct replay batches kept resident in HBM")
    ap.add_argument("--graph", action="store_true", help="also measure the opt-in whole-step HIP-graph replay as a secondary leg (eager vs replay; "
                    "profiles/BENCH_r06_run*.json were taken with it). Off by default: the runtime's graph path is erratic once other streams / "
                    "contexts exist (DESIGN §2) and a secondary leg must not be able to endanger the line")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for the profile targets of tools/gpurun.sh)")
    ap.add_argument("--dry", action="store_true", help="no GPU: dry contexts + gloo (launcher / sharding / merge / JSON contract only)")
    args = ap.parse_args()
    maybe_self_launch(args)

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` and let it launch its own ranks)")
    dry = args.dry
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no GPU is visible (use --dry for the launcher-only check)")
    import torch.distributed as dist
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from celebbasis_amd import build as cb_build
    if rank == 0:
        cb_build.build(verbose=False)
    if world > 1:
        dist.barrier()
    from celebbasis_amd import lib, synthetic, sharding

    # ---- seeded synthetic weights at the true SD v1-4 shapes (no checkpoints offline), same on every rank
    shapes = synthetic.trainer_shapes()
    state = shapes if dry else synthetic.synthetic_state(shapes, seed=0)
    ctx = lib.Context(device=-1) if dry else lib.Context()
    tr = lib.Trainer(ctx, lib.WeightTable(state))
    n_params = sum(int(torch.Size(v).numel()) if dry else v.numel() for v in state.values())

    identity = rank % 10            # the dense identity slot this rank personalises (max_ids = 10)
    batches = [None if dry else synthetic.synthetic_batch(1, step=s, device=dev, identity=identity, seed=23 + rank) for s in range(args.batches)]
    lr = 5e-3       # base_learning_rate 5e-3 * bs 1 * ngpu 1 * accum 1 (main_id_embed.py:767-786)

    def run(i):
        if dry:
            tr.step(1, None, None, None, None, None, None, lr)
            return
        b = batches[i % len(batches)]
        tr.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], lr)

    only3 = args.trace_leg == "config3"
    elapsed = 1.0 if only3 else timed_loop(run, args.steps, args.warmup, world, dist, dev, dry)
    loss = None if (dry or only3) else float(tr.get_state("loss")[0])

    # ---- secondary measurement (rank 0, N == 1; never `value`): the same K steps through the opt-in whole-step HIP-graph replay
    # (cb_trainer_set_graph_replay; bitwise identical results) — wall clock and the HOST thread's CPU time per step next to the eager
    # path's. Runs LAST among the timed legs: the runtime's graph-internal streams change the hardware-queue mapping of later streams.
    def graph_leg():
        if not (rank == 0 and world == 1 and not dry and not only3 and args.graph and not args.no_graph):
            return None
        res = {}
        for tag, on in (("eager", False), ("graph", True)):
            tr.set_graph_replay(on)
            for i in range(3):
                run(i)
            torch.cuda.synchronize(); tw, tc = time.perf_counter(), time.thread_time()
            for i in range(args.steps):
                run(3 + i)
            cpu = time.thread_time() - tc
            torch.cuda.synchronize(); el = time.perf_counter() - tw
            res[tag] = dict(steps_per_s=round(args.steps / el, 4), ms_per_step=round(el / args.steps * 1e3, 3), host_cpu_ms_per_step=round(cpu / args.steps * 1e3, 3))
        res.update(tr.graph_stats())
        tr.set_graph_replay(False)
        res["note"] = "opt-in cb_trainer_set_graph_replay: one captured graph of the step's ~1 650 launches replayed per step; not used for `value`"
        return res

    # ---- secondary measurement (rank 0, not the headline `value`): the same K steps with the first stage of batch i+1
    # (VAE encode + FR backbone, no gradients) prefetched on its own stream under the UNet of batch i (cb_trainer_prefetch).
    # Every timed iteration still executes one VAE encode, one FR pass and one full step; results are bitwise identical.
    pipelined = None
    if rank == 0 and world == 1 and not args.no_pipelined and not dry and not only3:      # N == 1 only, like the CPU baseline: keeps the N > 1 runs lean
        def nb(i):
            return batches[i % len(batches)]
        b0 = nb(0)
        tr.prefetch(b0["image"], b0["faces"], b0["post_noise"])
        for i in range(2):                                   # untimed: fill the pipeline
            n = nb(i + 1); tr.prefetch(n["image"], n["faces"], n["post_noise"]); run(i)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(2, 2 + args.steps):
            n = nb(i + 1); tr.prefetch(n["image"], n["faces"], n["post_noise"]); run(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - tp
        run(2 + args.steps)      # untimed: consume the outstanding prefetch, so that no later step (the profiled one) finds its first stage done
        torch.cuda.synchronize()
        pipelined = dict(steps_per_s=round(args.steps / el, 4), ms_per_step=round(el / args.steps * 1e3, 3),
                         note="first stage of batch i+1 overlapped with step i (opt-in cb_trainer_prefetch); not used for `value`")

    # ---- checkpoint merge (outside the timed region): the one collective of the job. Every rank contributes the coefficient block
    # of its own identity slot; rank 0 must end up with exactly `world` non-empty slots.
    if dry:
        coef = torch.zeros(10, 1024); coef[identity] = 1.0 + rank
    else:
        coef = tr.get_state("id_coefficients").reshape(10, -1)
    merged = sharding.gather_identity_blocks([identity], coef[identity:identity + 1], 10, device=dev)
    merged_identities = int((merged.abs().sum(dim=1) > 0).sum().item())
    assert merged_identities == min(world, 10), f"checkpoint merge returned {merged_identities} identities for {world} ranks"

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        if dry or only3:
            roofline, phases, st = None, {}, ctx.stats()
        else:
            roofline, phases, st = profile_step(ctx, run, PEAK_FP32_MFMA_TFLOPS)
            roofline["traffic"], roofline["traffic_detail"] = pmc_traffic(roofline["kernel"])
            roofline["frac_range_boxes"] = frac_range(roofline["frac"])
        out = {
            "metric": "personalization train steps/sec @ SD v1-4 512x512 bs=1", "value": round(value, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: single identity, SD v1-4 512x512, bs=1, fp32, full step "
                                   "(VAE encode + FR-net + CLIP fwd/bwd + UNet fwd/bwd + AdamW)",
                       "identities_per_gpu": 1, "global_batch": world, "latent": "4x64x64", "tokens": 77,
                       "parallelism": f"identity-sharded x{world} (no data-path collective)", "params_frozen": int(n_params),
                       "trainable": 525312, "kernel_launches_per_step": st["launches"]},
            "identities_per_sec": round(value / STEPS_PER_IDENTITY, 6),
            "step_tflop": round(st["gemm_flops"] / 1e12, 4),
            "step_tflops_per_gpu": round(st["gemm_flops"] / 1e12 / (ms_per_step * 1e-3), 2),
            "loss": loss,
            "phase_ms": phases,
            "graph_replay": None,
            "pipelined": pipelined,
            "merged_identities": merged_identities,
            "per_rank": list(PER_RANK) or None,
            "build": tree_stamp(),
            "roofline": roofline,
        }
        if dry:
            out["dry"] = True
        if only3:
            out["trace_leg"] = "config3"; out["value"] = None; out["ms_per_step"] = None
    # ---- secondary: BASELINE config 3 (bf16 UNet / VAE + fp32 coefficient gradient, bs=4) on its own context, N == 1 only
    if rank == 0 and world == 1 and not args.no_config3 and not dry:
        B3 = 4
        ctx3 = lib.Context(dtype="bf16")
        tr3 = lib.Trainer(ctx3, lib.WeightTable(state))
        b3 = [synthetic.synthetic_batch(B3, step=s, device=dev, identity=identity, seed=523) for s in range(2)]
        lr3 = 5e-3 * B3

        def run3(i):
            b = b3[i % len(b3)]
            tr3.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], lr3)
        el3 = timed_loop(run3, args.steps, args.warmup, 1, dist, dev, False)
        rf3, ph3, st3 = profile_step(ctx3, run3, PEAK_BF16_MFMA_TFLOPS, bf16=True)
        # the same K steps with the gradient-free first stage of batch i+1 (VAE encode + FR backbone) enqueued on the prefetch stream
        # under the UNet of batch i (cb_trainer_prefetch, `celebbasis_amd.main --pipeline`): every iteration still runs one
        # VAE encode, one FR pass, one CLIP forward and one full UNet step; results are bitwise identical. Secondary, like `pipelined`.
        pipe3 = None
        if not args.no_pipelined:
            nb3 = lambda i: b3[i % len(b3)]
            tr3.prefetch(nb3(0)["image"], nb3(0)["faces"], nb3(0)["post_noise"])
            for i in range(2):
                n = nb3(i + 1); tr3.prefetch(n["image"], n["faces"], n["post_noise"]); run3(i)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(2, 2 + args.steps):
                n = nb3(i + 1); tr3.prefetch(n["image"], n["faces"], n["post_noise"]); run3(i)
            torch.cuda.synchronize()
            elp = time.perf_counter() - tp
            run3(2 + args.steps)      # untimed: consume the outstanding prefetch
            torch.cuda.synchronize()
            pipe3 = dict(samples_per_s=round(B3 * args.steps / elp, 3), ms_per_step=round(elp / args.steps * 1e3, 3),
                         note="first stage of batch i+1 overlapped with step i (cb_trainer_prefetch); secondary figure")
        rf3["traffic"], rf3["traffic_detail"] = pmc_traffic(rf3["kernel"], bf16=True)
        rf3["frac_range_boxes"] = frac_range(rf3["frac"], bf16=True)
        ms3 = el3 / args.steps * 1e3
        out["config3"] = {"workload": "BASELINE config 3: single identity, SD v1-4 512x512, bs=4, bf16 UNet / VAE (storage + MFMA), fp32 CLIP / "
                                      "coefficient head / gradient / AdamW, full step", "dtype": "bf16", "batch": B3,
                          "steps_per_s": round(args.steps / el3, 4), "samples_per_s": round(B3 * args.steps / el3, 3), "ms_per_step": round(ms3, 3),
                          "vs_config2_samples_per_s": None if only3 else round(B3 * args.steps / el3 / out["value"], 3),
                          "step_tflop": round(st3["gemm_flops"] / 1e12, 4), "step_tflops_per_gpu": round(st3["gemm_flops"] / 1e12 / (ms3 * 1e-3), 2),
                          "kernel_launches_per_step": st3["launches"], "loss": float(tr3.get_state("loss")[0]), "phase_ms": ph3, "pipelined": pipe3,
                          "roofline": rf3}
        tr3.close(); ctx3.close()
    # ---- CPU baseline (rank 0, N == 1 only): the oracle on this box's host cores, bounded sample
    if out is not None:
        try:                                   # a secondary leg must never take the line down with it
            out["graph_replay"] = graph_leg()
        except Exception as e:                 # noqa: BLE001
            out["graph_replay"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            try:
                tr.set_graph_replay(False)
            except Exception:                  # noqa: BLE001
                pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        from oracle import cb_oracle as O
        cores = min(args.cpu_threads, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        cpu_state = {k: v for k, v in state.items()}
        st_cpu = O.TrainState(cpu_state)
        times = []
        for s in range(args.cpu_steps):
            b = {k: v.cpu() for k, v in batches[s % len(batches)].items()}
            c0 = time.perf_counter()
            O.train_step(cpu_state, st_cpu, b, lr)
            times.append(time.perf_counter() - c0)
        best = min(times)
        out["cpu_baseline"] = {"value": round(1.0 / best, 5), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{args.cpu_steps} full steps of the same workload (oracle/cb_oracle.train_step, torch-CPU fp32), "
                                         f"best of {args.cpu_steps}: {best:.2f} s/step"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
