#!/usr/bin/env python
"""probe_temporaries.py — root-cause probe for the "temporary freed right after an asynchronous native call" hazard
(VERDICT r1 weak #11): which stream reads a caller buffer after the call has returned, and does PyTorch's caching allocator
know about it?

Experiment: a mini trainer step is enqueued with freshly created device TEMPORARIES that are dropped right after the call, and
the very next statement creates new tensors of the same sizes (the caching allocator hands the same blocks back) and fills them
with garbage from pageable host memory. Result = loss / dW of the step. Variants:
  keep      the temporaries stay referenced until a sync            (reference result)
  drop      dropped + blocks refilled with garbage                  (race if a library-owned stream still reads them)
  record    dropped, but every input was `record_stream`-ed on the library's side stream (cb_ctx_aux_stream) first
Each variant runs on PyTorch's default stream and on a non-default stream.

Second experiment — cb_trainer_prefetch, the one entry point that keeps reading caller memory on a library stream AFTER it has
returned (until the consuming cb_trainer_step is enqueued):
  pf_keep      prefetch(batch); step(batch) with the tensors referenced throughout          (reference result)
  pf_drop_raw  raw cb_trainer_prefetch, then image / faces / post_noise are dropped and their blocks refilled with garbage
               BEFORE the step (which consumes the prefetched result by pointer identity)  -> the race, if there is one
  pf_drop_rec  the same with lib.Trainer.prefetch, which record_stream()s the three tensors on cb_trainer_prefetch_stream
Writes gpurun_out/probe_temporaries.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from celebbasis_amd import lib
    from oracle import cb_oracle as O
    from test_gpu_models import MINI, make_batch, trainer_cfg
    dev = torch.device("cuda:0")
    m = MINI
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=1)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    out = {}
    for sname in ("default", "side"):
        stream = torch.cuda.default_stream(dev) if sname == "default" else torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):
            ctx = lib.Context()
            aux = ctx.aux_stream() if hasattr(ctx, "aux_stream") else None
            for variant in ("keep", "drop", "record"):
                if variant == "record" and aux is None:
                    continue
                bad = 0
                ref = None
                for rep in range(12):
                    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
                    batch = make_batch(2, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100, m["max_ids"])
                    d = {k: v.contiguous().to(dev) for k, v in batch.items()}
                    if variant == "record":
                        for v in d.values():
                            v.record_stream(aux)
                    tr.step(d["image"], d["faces"], d["ids"], d["tokens"], d["t"], d["noise"], d["post_noise"], 1e-2)
                    if variant != "keep":
                        shapes_ = {k: (v.shape, v.dtype) for k, v in d.items()}
                        del d
                        junk = [torch.full(s, 7, dtype=dt).to(dev) for s, dt in shapes_.values()]      # same-size blocks, garbage
                    ctx.sync()
                    res = (float(tr.get_state("loss")[0]), float(tr.get_state("mlp.grad_weight").abs().sum()))
                    tr.close()
                    if ref is None:
                        ref = res
                    bad += int(res != ref) if variant == "keep" else 0
                    out.setdefault(f"{sname}.{variant}", []).append(res)
            # ---- prefetch: the out-of-stream-order reader
            import ctypes as C
            L = lib.load()
            for variant in ("pf_keep", "pf_drop_raw", "pf_drop_rec"):
                for rep in range(12):
                    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
                    batch = make_batch(2, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100, m["max_ids"])
                    d = {k: v.contiguous().to(dev) for k, v in batch.items()}
                    b = lib.Batch(); b.B = 2
                    b.image, b.faces, b.post_noise = lib.ptr(d["image"]), lib.ptr(d["faces"]), lib.ptr(d["post_noise"])
                    b.ids, b.tokens, b.t = lib.ptr(d["ids"], torch.int64), lib.ptr(d["tokens"], torch.int64), lib.ptr(d["t"], torch.int64)
                    b.noise = lib.ptr(d["noise"])
                    # keep the main stream busy so that the prefetch (which waits for it) starts late
                    busy = torch.randn(4096, 4096, device=dev)
                    for _ in range(6):
                        busy = busy @ busy * 1e-3
                    if variant == "pf_drop_rec":
                        tr.prefetch(d["image"], d["faces"], d["post_noise"])
                    else:
                        lib.check(L.cb_trainer_prefetch(tr.h, C.byref(b)))
                    if variant != "pf_keep":
                        shp = [(d[k].shape, d[k].dtype) for k in ("image", "faces", "post_noise")]
                        del d["image"], d["faces"], d["post_noise"]
                        junk = [torch.full(sh, 7, dtype=dt, device=dev) for sh, dt in shp]     # same-size blocks, refilled on the main stream
                    lib.check(L.cb_trainer_step(tr.h, C.byref(b), C.c_float(1e-2)))
                    ctx.sync()
                    res = (float(tr.get_state("loss")[0]), float(tr.get_state("mlp.grad_weight").abs().sum()))
                    tr.close()
                    out.setdefault(f"{sname}.{variant}", []).append(res)
            ctx.close()
    summary = {}
    for sname in ("default", "side"):
        base = out[f"{sname}.keep"][0]
        for variant in ("keep", "drop", "record", "pf_keep", "pf_drop_raw", "pf_drop_rec"):
            rs = out.get(f"{sname}.{variant}")
            if rs:
                summary[f"{sname}.{variant}"] = dict(mismatches=sum(int(r != base) for r in rs), runs=len(rs), first=rs[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "probe_temporaries.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
