#!/usr/bin/env python
"""tools/ab_step.py — round 6: within-process A/B of the config-2 step (SD v1-4, 512x512, bs=1, fp32; `--bf16` = config 3, bs=4)
under the two launch-structure switches: the deferred step tail (cb_trainer_set_tail_overlap) and the in-launch split-K reduction
(cb_debug_splitk_inlaunch). Interleaved rounds (the variants alternate, so box drift hits all of them alike), wall clock per step
and the HOST time of one `step` call on an idle queue (= what the host needs to enqueue a step).

    python tools/ab_step.py [--steps 15] [--rounds 3] [--bf16]        -> one JSON line (also gpurun_out/ab_step.json)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--only", action="store_true", help="the default configuration only (for process-level A/B: CB_PLAN_FILE, CB_* switches)")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    from celebbasis_amd import build as cb_build
    cb_build.build(verbose=False)
    from celebbasis_amd import lib, synthetic
    dev = torch.device("cuda", 0)
    B = 4 if args.bf16 else 1
    state = synthetic.synthetic_state(synthetic.trainer_shapes(), seed=0)
    ctx = lib.Context(dtype="bf16" if args.bf16 else "f32")
    tr = lib.Trainer(ctx, lib.WeightTable(state))
    del state
    batches = [synthetic.synthetic_batch(B, step=s, device=dev, identity=0, seed=23) for s in range(4)]
    lr = 5e-3 * B

    def run(i):
        b = batches[i % 4]
        tr.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], lr)

    L = lib.load()
    variants = [("tail+inlaunch", 1, 1), ("tail only", 1, 0), ("inlaunch only", 0, 1), ("neither (round 5)", 0, 0)]
    if args.only:
        variants = variants[:1]
    res = {v[0]: [] for v in variants}
    host = {v[0]: [] for v in variants}
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for name, tail, inl in variants:
            tr.set_tail_overlap(bool(tail))
            lib.check(L.cb_debug_splitk_inlaunch(inl))
            for i in range(2):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(0)
            host[name].append((time.perf_counter() - t0) * 1e3)          # enqueue time of one step on an idle queue
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                run(i)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / args.steps * 1e3)
    lib.check(L.cb_debug_splitk_inlaunch(1)); tr.set_tail_overlap(True)
    out = dict(tag=args.tag, plan_file=os.environ.get("CB_PLAN_FILE"), workload="config 3 (bf16, bs=4)" if args.bf16 else "config 2 (fp32, bs=1)", steps=args.steps, rounds=args.rounds,
               ms_per_step={k: [round(x, 3) for x in v] for k, v in res.items()},
               ms_per_step_best={k: round(min(v), 3) for k, v in res.items()},
               host_enqueue_ms={k: [round(x, 2) for x in v] for k, v in host.items()},
               loss=float(tr.get_state("loss")[0]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", ("ab_step_bf16" if args.bf16 else "ab_step") + (("_" + args.tag) if args.tag else "") + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
