#!/usr/bin/env python
"""bench.py — images/s of the Mask-RCNN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of MaskRCNN.predict over a batch of 8 synthetic 1024×1024×3 uint8 images that are
already resident in HBM (BASELINE.json configs[1]: ResNet101+FPN, 1024², batch 8 per GPU, pre_nms 6000,
max_proposals 1000), followed — when N > 1 — by the one collective the path has: an RCCL all-gather of
the fixed-size detection records.  Weak scaling: every GPU processes its own 8 images.
Weights are seeded synthetic weights of the exact architecture ("forced full load": 1000 proposals and
100 detections per image, so no data-dependent stage idles); there is no network for checkpoints.

Default compute mode: the one a drop-in host gets (round 6).  Rank 0 writes the synthetic artefacts, calibrates them once the way
`convert --calibrate` does (convert.calibrate_artefact: the split exponents of the tensor groups stored in MaskRCNN.mrcw) and every rank
loads the model WITHOUT naming a precision (MRCNN_DEFAULT, as `MaskRCNN()` in ViewController.swift:37): an artefact that carries stored
exponents resolves to f32x3 (`config.model_loaded_by`, `dtype` in the line).  `--dtype f32x3` etc. load explicitly as before.
f32x3 — fp32 tensors, fp32 accumulation, products formed on the fp16 matrix cores from a three-part
fp16 split of the activation × the fp16-stored filter: an activation with 0.5 <= |a| < 65504 is carried EXACTLY (all 24
significand bits, so a*w is the exact product), a smaller one to 2^-25 ABSOLUTE (rounded to nearest: the third part
reaches the fp16 subnormal step) — so the engine stores every tensor a split convolution reads as 2^e * value, with a power-of-two
exponent per tensor group chosen by ONE calibration predict (round 4: mrcnn_model_calibrate_split; max |a| * 2^e in [2^11, 2^12),
folded into the layers' scale / shift at no run-time cost): fp32-grade at any activation scale (tests/test_gpu_split_scale.py;
the un-prescaled curve: profiles/r03_split_scale_curve.txt).  bench.py calibrates on the first images of the seed-1 stream
(`split` in the line; --no-calibrate = every exponent 0, the round-3 behaviour).  Against an fp64 evaluation of the same graph the
mode is CLOSER than the fp32-MFMA engine (profiles/r03_fp64_trunk_parity.json); end to end it agrees with the CPU oracle on
1600 / 1600 detections of the 16 images of parity_e2e below and on 25 593 / 25 600 (99.97 %, 249 / 256 images fully matched) of
profiles/r03_parity_e2e_256.json — the residue is near-tie ordering between two fp32 evaluations that sum in different orders
(the exact-fp32 engine misses more).  The fp32-MFMA mode (`--dtype f32`) is timed under other_modes.

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks ITSELF (re-executes this file under
`python -m torch.distributed.run --standalone --nproc-per-node N`, rank 0's JSON line is the output); a --gpus / WORLD_SIZE
mismatch, or fewer than N visible GPUs, is an error — never a silent n_gpus: 1 line.  At N > 1 the all-gather of step i
runs on its own stream under the predict of step i + 1 (mrcnn_dist_all_gather_records_async); the timed region ends with
the last exchange joined.  The process holds ONE RCCL user: the native communicator of mrcnn_dist_* (it binds the RCCL copy the
process already mapped); torch.distributed runs over gloo and carries host-side rendezvous, barriers and the max-over-ranks
only.  Rank 0 writes the synthetic model directory once; `per_rank_ms_per_step` carries every rank's own time.

Rank 0 prints ONE JSON line with, besides the contract fields:
  roofline     — dominant conv kernel (the tile class with the largest share of the step): its ALGORITHMIC flops per
                 launch ÷ its average launch duration, both measured live with HIP events on the launching stream
                 in --event-steps further steps of the same batch run RIGHT BEHIND the timed region (round 6: every timed
                 step is uninstrumented — bracketing every launch drains the queue between kernels and cost ~2 % of an
                 fp32 step, ~13 % of an fp16 one; shares are taken against those instrumented steps' own wall time); every
                 tile class carries its bound (mfma | hbm: the larger of flops / MFMA peak and ALGORITHMIC bytes / 8 TB/s),
                 GB/s and frac_of_hbm beside TFLOP/s — the 1x1 layers of the fp32-tensor modes are HBM-bound; peak = dense fp16 MFMA 2500 TFLOP/s ÷ the MFMA passes per product
                 (f32x3: 3, f32s: 2, f16: 1) or 157.3 TFLOP/s for the fp32-MFMA mode
  cpu_baseline — the oracle (torch-CPU fp32 network + the C restatement of the custom layers) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N = 1 only), to the reference's
                 protocol: 1 warm-up + 5 timed images
  parity_e2e   — HIP predict vs the oracle's predict on 16 images, per compute mode: share of detections with the
                 same class id and a box within 1e-4
  gpu_busy     — GPU seconds of the predicts inside the timed region, from HIP events on the model's stream (one pair per
                 predict), next to the wall clock of the region: measured in THIS run
  h2d_included — the same step with the batch in pinned HOST memory (25 MB H2D + 2.5 MB D2H per step inside the timing):
                 what the reference's per-image timing (EvaluateCommand.swift:167-179) would include; never `value`
  profiles_ref — constants read from committed files under profiles/ (PMC traffic, the matrix cores' sustained rate):
                 NOT measured in this run, kept apart from the live numbers for that reason
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: bf16/f16 MFMA dense peak (not the 2:1-sparse figure)
GFLOP_PER_IMAGE_SURVEY = 777.3         # SURVEY.md §8(d), R101 1024² 81 classes
# C1..C5 at 1024²: the sum of SURVEY.md §8(d)'s per-stage figures (C1 4.93, C2 27.92, C3 39.73, C4 213.14 | 57.98, C5 30.60).  BASELINE.md §2's
# column "of which backbone C1-C5" prints 344.9 / 189.8 — that is this sum PLUS the box head's 28.62 (a slip in the table, kept there unedited).
GFLOP_BACKBONE_SURVEY = {"resnet101": 316.32, "resnet50": 161.16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--num-classes", type=int, default=81, help="BASELINE configs[4] uses 2")
    ap.add_argument("--pre-nms", type=int, default=6000, help="preNMSMaxProposals (BASELINE configs[4]: 12000)")
    ap.add_argument("--dtype", default="default", choices=["default", "f32", "f16", "f32s", "f32x3"],
                    help="compute mode of the convolutions.  default: the artefact is calibrated once (convert.calibrate_artefact = `convert --calibrate`) and loaded with "
                         "MRCNN_DEFAULT, which resolves to f32x3 — what a drop-in host gets.  f32x3: fp32 tensors, products a*w formed on the "
                         "fp16 matrix cores from a three-part split of the fp32 activation against the fp16-stored filter "
                         "(task.py:90; exact for 0.5 <= |a| < 65504, the activation carried to 2^-25 absolute below), fp32 accumulate — measured closer to an fp64 evaluation than the fp32-MFMA engine "
                         "(profiles/r03_fp64_trunk_parity.json), 99.97 %% of 25 600 detections matched end to end with the CPU oracle (profiles/r03_parity_e2e_256.json); "
                         "f32: v_mfma_f32_32x32x2_f32 (round-1 headline, now under other_modes); f32s: two-part split; "
                         "f16: fp16 tensors + fp16 MFMA (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-probe", action="store_true", help="skip the in-process MFMA probe and the clock / power sampler (roofline.sustained_peak_live, clock_mhz)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="split modes: skip the calibration predict (every split exponent 0: the round-3 behaviour)")
    ap.add_argument("--cpu-images", type=int, default=5,
                    help="timed oracle images after 1 warm-up (SURVEY.md §8d / EvaluateCommand.swift:165: 5 images)")
    ap.add_argument("--e2e-images", type=int, default=16,
                    help="images on which HIP predict is compared end to end with the CPU oracle's predict (parity_e2e); the "
                         "warm-up and timed images of the cpu_baseline leg are the first of them; 0 = skip")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the all-gather leg even at world size 1 (self-test of the N > 1 path)")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the short extra timed loops of the other compute modes (reported under other_modes, N = 1 only)")
    ap.add_argument("--event-steps", type=int, default=3,
                    help="conv launches are bracketed by HIP events during N further steps behind the timed region "
                         "(the events cost ~2 %% of an fp32 step, ~13 %% of an fp16 one: kept out of `value`); 0 = as many as --steps")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: refusing to print a line for a "
                         f"different number of GPUs than asked for\n")
        sys.exit(2)

    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL's version banner on the
    # first communicator) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible\n")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # ONE RCCL user per process: the data-path collective is the native ncclAllGather behind the C ABI
        # (mrcnn_dist_*); the torch process group only carries the 128-byte rendezvous id, the model directory's path, the
        # barriers and the max-over-ranks of the elapsed time — host-side work, so it runs over gloo and holds no communicator
        # on the GPU.  (mrcnn_dist_* binds the RCCL copy the process has already mapped — torch's — before loading its own.)
        host_group_init(rank, world)

    pkg = importlib.import_module("mask-rcnn-coreml_amd")
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")

    cfg = pkg.ModelConfig(architecture=args.arch, input_image_shape=(args.size, args.size, 3), num_classes=args.num_classes,
                          pre_nms_max_proposals=args.pre_nms)
    # rank 0 writes the 250 MB synthetic model directory ONCE; the other ranks of the node load the same files
    B = args.batch
    # (the calibration images: their OWN stream, seed 7 — disjoint from the timed batch and from the end-to-end images, which come from the seed-1 stream)
    calib_np = np.random.default_rng(7).integers(0, 256, (min(B, 2), args.size, args.size, 3), dtype=np.uint8)
    by_default = args.dtype == "default"
    stored_info = {}

    def write_artefacts(d):
        weights.save_synthetic_models(d, cfg, seed=0, forced_load=True)
        if by_default and not args.no_calibrate:
            # what `python -m mask-rcnn-coreml_amd.convert ... --calibrate images` does behind the conversion: one calibration predict on the GPU,
            # the exponent vector stored in MaskRCNN.mrcw ("split_exp.<group>")
            convert = importlib.import_module("mask-rcnn-coreml_amd.convert")
            stored_info.update(convert.calibrate_artefact(d, calib_np, verbose=False))

    model_dir = shared_model_dir(rank, use_dist, write_artefacts)
    # by default NO precision is named (MRCNN_DEFAULT): the library resolves it from the artefact — stored exponents -> f32x3
    m = models.load_maskrcnn(model_dir, max_batch=args.batch) if by_default else models.load_maskrcnn(model_dir, max_batch=args.batch, compute_dtype=args.dtype)
    args.dtype = m.compute_dtype
    loaded_by = (f"MRCNN_DEFAULT (no precision named, as MaskRCNN() in ViewController.swift:37) -> {m.compute_dtype}: "
                 + ("the artefact carries stored split exponents (convert.calibrate_artefact = convert --calibrate)" if m.get_int("split_exponents_from_artefact")
                    else "the artefact carries no stored split exponents")) if by_default else f"explicit compute_dtype {args.dtype}"
    # Scale-aware split (include/maskrcnn_hip.h: mrcnn_model_calibrate_split): the split modes run with a power-of-two pre-scale
    # per tensor group, chosen from ONE calibration predict on a canonical batch — two images of a separate seeded stream, the
    # same on every rank, so every rank holds the same exponent vector (per-image results do not depend on the rank).
    split_info = None
    calib = torch.from_numpy(calib_np).to(dev)
    if by_default:
        split_info = dict(stored_info, source="stored in MaskRCNN.mrcw by convert.calibrate_artefact, applied by mrcnn_model_load") if stored_info else None
    elif args.dtype in ("f32x3", "f32s") and not args.no_calibrate:
        split_info = m.calibrate_split(calib)
    # synthetic batch, uint8 uniform[0,255], seed 1; a different slice of the stream per rank
    rng = np.random.default_rng(1)
    rng.bit_generator.advance(rank * B * args.size * args.size * 3)
    images = torch.from_numpy(rng.integers(0, 256, (B, args.size, args.size, 3), dtype=np.uint8)).to(dev)
    det = torch.empty((B, m.max_detections, 6), dtype=torch.float32, device=dev)
    mask = torch.empty((B, m.max_detections, m.mask_size, m.mask_size), dtype=torch.float32, device=dev)
    # N > 1: the exchange goes through the C ABI (mrcnn_dist_*: ncclAllGather from librccl on the model's stream) — the
    # shipped multi-GPU path; torch.distributed only carries the 128-byte rendezvous id, the barriers and the timing reduce.
    gather = None
    if use_dist:
        gather = dmod.NativeDist(rank, world, broadcast_id(rank, dmod.NativeDist.unique_id if rank == 0 else None))
        all_det = torch.empty((world * B, m.max_detections, 6), dtype=torch.float32, device=dev)
        all_mask = torch.empty((world * B, m.max_detections, m.mask_size, m.mask_size), dtype=torch.float32, device=dev)

    def step():
        m.predict_into(images, det, mask, sync=True)       # returns after the model's stream has drained
        if gather is not None:
            # the exchange of this step runs on the handle's own stream, under the NEXT step's predict; the previous one
            # is joined first (its outputs are about to be overwritten, and a failure on any rank surfaces here)
            gather.wait()
            gather.all_gather_records_async(m, det, mask, world * B, all_det, all_mask)

    def drain():
        if gather is not None:
            gather.wait()

    for _ in range(args.warmup):
        step()
    drain()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    busy0, calls0 = m.get_int("gpu_busy_us"), m.get_int("predict_calls")
    t0 = time.perf_counter()
    for i in range(args.steps):                        # EXACTLY K uninstrumented steps: no stage timer, no event around any launch
        step()
    drain()                                            # the last exchange belongs to the timed region
    fence()
    elapsed = time.perf_counter() - t0
    busy_s = (m.get_int("gpu_busy_us") - busy0) * 1e-6
    busy_calls = m.get_int("predict_calls") - calls0
    # The instrumented steps: the SAME step a few more times right behind the timed region, with every conv launch
    # bracketed by HIP events and the stage timer on — `roofline` and `stage_ms_last_step` come from these, `value` from none of them.
    ev_steps = 0 if args.no_kernel_events else (args.event_steps if args.event_steps > 0 else args.steps)
    m.enable_timing(True)
    if not args.no_kernel_events:
        m.conv_profile_enable(True)
    torch.cuda.synchronize()
    t_ev = time.perf_counter()
    for i in range(max(ev_steps, 1)):
        step()
    drain()
    torch.cuda.synchronize()
    ev_elapsed = time.perf_counter() - t_ev            # wall time of the instrumented steps: what the shares below are taken against
    if not args.no_kernel_events:
        m.conv_profile_enable(False)                   # window closed, totals kept
    if use_dist:
        dist.barrier()
    every = gather_elapsed(elapsed, world) if use_dist else [elapsed]
    per_rank_ms = [1e3 * e / args.steps for e in every]
    elapsed = max(every)                                        # the job is as slow as its slowest rank

    prof = m.conv_profile() if not args.no_kernel_events else None
    prof_groups = m.conv_profile_groups() if not args.no_kernel_events else None
    prof_bytes = m.conv_profile_bytes() if not args.no_kernel_events else None
    stages = m.stage_ms()
    n_prop = int(m.read_tensor("keep_count", 0)[0])
    n_det = int((det[0, :, 5] > 0).sum().item())
    # Same-run box normaliser, OUTSIDE the timed region (a sampler thread beside the timed loop cost 16 % of it: the
    # library's queries contend with the launches — gpu_busy fell to 0.84): right behind it the same step runs on for half a second, untimed,
    # while a host thread samples the shader clock / socket power through librocm_smi64; then 1.5 s of back-to-back MFMAs on changing
    # register operands, no operand traffic (mrcnn_bench_mfma_probe), give what the matrix cores of THIS box sustain right now.
    smi, live_probe = None, None
    if rank == 0 and not args.no_live_probe:
        m.enable_timing(False)
        sampler = SmiSampler(local_rank)
        sampler.start()
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.5:
            m.predict_into(images, det, mask, sync=True)
        smi = sampler.stop()
        live_probe = mfma_probe_live(1.5, args.dtype)

    if rank == 0:
        total_images = n_gpus * B * args.steps
        value = total_images / elapsed
        out = {
            "metric": "images/sec (1024x1024, COCO-80)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "dtype_note": {"f32x3": "fp32 tensors and accumulation; products a*w formed on the fp16 MFMA from a 3-part split of the activation "
                                    "(the filters are fp16 in the artefact, task.py:90): exact for 0.5 <= |a| < 65504, the activation "
                                    "carried to 2^-25 absolute (rounded to nearest) below — not scale-invariant like fp32; curve: "
                                    "profiles/r03_split_scale_curve.txt",
                           "f32": "fp32 tensors, v_mfma_f32_32x32x2_f32", "f32s": "fp32 tensors, 2-part split of the activations (23 of 24 bits, nearest rounding)",
                           "f16": "fp16 tensors, fp16 MFMA, fp32 accumulate; box path and outputs fp32"}[args.dtype],
            "config": {"workload": (f"BASELINE configs[1]: " if (args.arch, args.size, args.num_classes, args.pre_nms, B) == ("resnet101", 1024, 81, 6000, 8) else "")
                                   + f"{args.arch}+FPN {args.size}x{args.size}, batch {B} per GPU, "
                                   f"{args.num_classes} classes, pre_nms {args.pre_nms}, max_proposals 1000, max_detections 100; "
                                   f"synthetic seeded weights (forced full load)",
                       "model_loaded_by": loaded_by,
                       "global_batch": n_gpus * B, "parallelism": f"dp{n_gpus}" if n_gpus > 1 else "single",
                       "proposals_kept_image0": n_prop, "detections_image0": n_det},
            "stage_ms_last_step": {k: round(v, 3) for k, v in stages.items()},
            "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "ranks": [round(v, 3) for v in per_rank_ms]},
            "gpu_busy": {"gpu_seconds": round(busy_s, 4), "wall_seconds": round(elapsed, 4), "frac": round(busy_s / elapsed, 4),
                         "predicts": int(busy_calls),
                         "how": "HIP events on the model's stream around every predict of the timed region (rank 0), this run"},
        }
        if split_info is not None:
            out["split"] = dict(split_info, note="scale-aware split: power-of-two exponent per tensor group from one calibration predict "
                                "(max |a| * 2^e in [2^11, 2^12)); small / inexact = non-zero stored inputs below 2^-8 of their tensor's maximum / "
                                "below 0.5 (carried to 2^-25 absolute = 2^-36 of the maximum), over the calibration batch")
        if use_dist:
            out["rccl"] = {"native_shares_process_copy": importlib.import_module("mask-rcnn-coreml_amd._lib").lib().mrcnn_dist_rccl_shared(),
                           "torch_process_group": "gloo (host-side rendezvous / barriers only)"}
        if prof is not None:
            # dominant kernel = the conv tile class with the largest share of the step
            dom = max(prof, key=lambda k: prof[k][1])
            launches, ms, flops = prof[dom]
            all_ms = sum(v[1] for v in prof.values())
            all_fl = sum(v[2] for v in prof.values())
            if launches:
                achieved = flops / (ms * 1e-3) / 1e12
                # the split modes execute 2 / 3 fp16 MFMA flops per algorithmic flop: their ceiling is 1/2, 1/3 of the fp16 peak
                parts = {"f32": 1, "f16": 1, "f32s": 2, "f32x3": 3}[args.dtype]
                peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_FP16_MFMA_TFLOPS / parts
                ktypes = {"f32": "float,float", "f16": "_Float16,_Float16", "f32s": "float,_Float16", "f32x3": "float,_Float16"}[args.dtype]
                tail = {"f32": "2,2", "f16": "2,2", "f32s": "3,2", "f32x3": "3,3"}[args.dtype]
                kname = {"128x128": f"k_conv_mfma_glds<{ktypes},128,1,2,4,2,{tail}>", "128x64": f"k_conv_mfma_glds<{ktypes},64,1,1,4,2,...>",
                         "128x32": f"k_conv_mfma_glds<{ktypes},32,1,1,4,1,...>", "128x128w4": f"k_conv_mfma_glds<{ktypes},128,1,4,4,1,{tail}>",
                         "128x256tail": f"k_conv_halo<{parts},2,false,false,2,3,TAIL> (3x3 + 1x1 + shortcut of a bottleneck block in one launch; opt-in)",
                         "bneck": "k_bneck_h<C> (an identity bottleneck block — 1x1, 3x3, 1x1 + shortcut — as one persistent launch, both mid tensors on chip; C = 64 / 128 / 256)",
                         "256x256pp": "k_conv_pp<0>", "128xNhalo": f"k_conv_halo<{parts},TN,HEAD,.,TM,MAXPC,.,WN> (persistent halo tiles: 128 x 256 / 128 x 128 / 64 x 128, 256 x 64 / 128 x 64 for 64 columns; instantiations <{parts},2,false>, <{parts},2,true> = fused RPN heads, <{parts},1,false>, <{parts},2,false,.,1,5,.,1> = C2; k_conv_halo_lat on under-filled grids)"}[dom]
                out["roofline"] = {
                    "kernel": kname, "bound": "mfma",
                    "achieved": round(achieved, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "flops_counted": "algorithmic (2*M*N*K of the convolution)" + (
                        f"; the kernel EXECUTES {parts} fp16 MFMA flops per algorithmic flop ({round(achieved * parts, 1)} of "
                        f"{PEAK_FP16_MFMA_TFLOPS} TFLOP/s), so the peak for algorithmic flops is 1/{parts} of the fp16 MFMA peak" if parts > 1 else ""),
                    "traffic": pmc_traffic(args.dtype, dom),
                    "traffic_note": "from profiles_ref (separate rocprofv3 --pmc passes of this command, committed): not of this run",
                    "launches_per_step": launches // ev_steps, "event_steps": ev_steps,
                    "avg_launch_ms": round(ms / launches, 4),
                    "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
                    "share_of_step_time": round(ms / (1e3 * ev_elapsed), 4),
                    "instrumented": {"steps": ev_steps, "ms_per_step": round(1e3 * ev_elapsed / max(ev_steps, 1), 3),
                                     "note": "run right behind the timed region; `value` / `ms_per_step` above come from uninstrumented steps only"},
                    "by_tile_class": {k: tile_class_entry(v, prof_bytes[k], ev_steps, ev_elapsed, peak) for k, v in prof.items() if v[0]},
                    "all_conv_kernels": {"tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2),
                                         "frac": round(all_fl / (all_ms * 1e-3) / 1e12 / peak, 4),
                                         "gflop_per_image": round(all_fl / (B * ev_steps) / 1e9, 2),
                                         "survey_gflop_per_image": GFLOP_PER_IMAGE_SURVEY,
                                         "share_of_step_time": round(all_ms / (1e3 * ev_elapsed), 4)},
                }
                if prof_groups and prof_groups["backbone"][0]:
                    # the subset north_star's ">= 50 % MFMA roofline for the backbone convs" is worded on: conv1 + res2..res5 (C1-C5)
                    bl, bms, bfl = prof_groups["backbone"]
                    out["roofline"]["backbone_convs"] = {
                        "tflops": round(bfl / (bms * 1e-3) / 1e12, 2), "frac": round(bfl / (bms * 1e-3) / 1e12 / peak, 4),
                        "gflop_per_image": round(bfl / (B * ev_steps) / 1e9, 2), "survey_gflop_per_image": GFLOP_BACKBONE_SURVEY.get(args.arch),
                        "launches_per_step": bl // ev_steps, "share_of_step_time": round(bms / (1e3 * ev_elapsed), 4),
                        "what": "conv1 + the res2..res5 stages (SURVEY.md section 8d 'backbone convs'); every other convolution (FPN, RPN, heads) is in all_conv_kernels only"}
                if live_probe:
                    # held against what THIS box's matrix cores sustain in this process, the figure is comparable across boxes of the pool
                    out["roofline"]["sustained_peak_live"] = {
                        "value": round(live_probe["tflops"] / parts, 1), "unit": "TFLOP/s", "probe_tflops_executed": round(live_probe["tflops"], 1),
                        "probe_mhz_equivalent": round(live_probe["mhz"], 0), "seconds": live_probe["seconds"],
                        "how": "mrcnn_bench_mfma_probe in this process right behind the timed loop: every wave on back-to-back "
                               + ("v_mfma_f32_32x32x2_f32" if args.dtype == "f32" else "v_mfma_f32_32x32x16_f16")
                               + ", register operands changing per instruction, no operand traffic" + (f"; divided by the {parts} MFMA passes per algorithmic flop" if parts > 1 else "")}
                    out["roofline"]["frac_of_live_sustained"] = round(achieved * parts / live_probe["tflops"], 4)
                    out["roofline"]["all_conv_kernels"]["frac_of_live_sustained"] = round(all_fl / (all_ms * 1e-3) / 1e12 * parts / live_probe["tflops"], 4)
                    if "backbone_convs" in out["roofline"]:
                        out["roofline"]["backbone_convs"]["frac_of_live_sustained"] = round(out["roofline"]["backbone_convs"]["tflops"] * parts / live_probe["tflops"], 4)
                if smi:
                    out["roofline"]["clock_mhz"] = smi.get("sclk_mhz_mean")
                    out["roofline"]["board"] = smi
                out["profiles_ref"] = {
                    "note": "constants read from committed files under profiles/ — NOT measured in this run",
                    "traffic_bytes_per_launch_dominant_kernel": pmc_traffic(args.dtype, dom), "traffic_source": pmc_traffic_source(args.dtype),
                    "sustained_peak": sustained_peak(args.dtype, parts, achieved)}
        # images of the end-to-end parity leg (the oracle sees the same ones): seed 1 stream, after this rank's bench batch
        n_e2e = 0 if (n_gpus != 1 or args.no_cpu_baseline) else max(args.e2e_images, 0)
        e2e_imgs = rng.integers(0, 256, (max(n_e2e, 1 + args.cpu_images), args.size, args.size, 3), dtype=np.uint8)
        e2e_pred = {}

        def hip_predict_all(model):
            dd, kk = [], []
            for i in range(0, n_e2e, B):
                d_, k_ = model.predict(e2e_imgs[i:min(i + B, n_e2e)])
                dd.append(d_); kk.append(k_)
            return np.concatenate(dd), np.concatenate(kk)

        if n_e2e:
            e2e_pred[args.dtype] = hip_predict_all(m)
        if n_gpus == 1 and not args.no_other_modes:
            # the same workload in the engine's other compute modes (same images, same weights): not the headline value
            out["other_modes"] = {}
            for mode in ("f32", "f32x3", "f32s", "f16"):
                if mode == args.dtype:
                    continue
                mm = models.load_maskrcnn(model_dir, max_batch=B, compute_dtype=mode)
                if mode in ("f32x3", "f32s") and not args.no_calibrate:
                    mm.calibrate_split(calib)
                if n_e2e:
                    e2e_pred[mode] = hip_predict_all(mm)
                n_om = 10
                for _ in range(2):
                    mm.predict_into(images, det, mask, sync=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_om):
                    mm.predict_into(images, det, mask, sync=True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n_om
                # ... and two instrumented steps behind its timed loop: the backbone subset of THIS mode against its own MFMA roof
                bb = None
                if not args.no_kernel_events:
                    mm.conv_profile_enable(True)
                    for _ in range(2):
                        mm.predict_into(images, det, mask, sync=True)
                    mm.conv_profile_enable(False)
                    g_ = mm.conv_profile_groups()
                    p_ = mm.conv_profile()
                    parts_ = {"f32": 1, "f16": 1, "f32s": 2, "f32x3": 3}[mode]
                    peak_ = PEAK_FP32_MFMA_TFLOPS if mode == "f32" else PEAK_FP16_MFMA_TFLOPS / parts_
                    if g_["backbone"][0]:
                        bt = g_["backbone"][2] / (g_["backbone"][1] * 1e-3) / 1e12
                        at = sum(v[2] for v in p_.values()) / (sum(v[1] for v in p_.values()) * 1e-3) / 1e12
                        bb = {"backbone_convs": {"tflops": round(bt, 1), "frac": round(bt / peak_, 4)},
                              "all_conv_kernels": {"tflops": round(at, 1), "frac": round(at / peak_, 4)}, "peak_tflops": round(peak_, 1)}
                out["other_modes"][mode] = {"value": round(B / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": n_om,
                                            "roofline": bb,
                                            "note": {"f32": "exact-fp32 MFMA", "f16": "fp16 tensors + fp16 MFMA (BASELINE configs[3])",
                                                     "f32s": "fp32 tensors, two fp16 MFMA passes over a hi/lo split of the activations "
                                                             "(fp32-grade: parity-tested at the fp32 tolerances)",
                                                     "f32x3": "fp32 tensors, three fp16 MFMA passes over a three-part split of the "
                                                              "activations (exact for 0.5 <= |a| < 65504, 2^-25 absolute below)"}[mode]}
                del mm
        if n_gpus == 1:
            # the same step with the batch in pinned host memory: H2D of the images and D2H of the records inside the timing
            himg = images.cpu().pin_memory()
            hdet = torch.empty(det.shape, dtype=torch.float32).pin_memory()
            hmask = torch.empty(mask.shape, dtype=torch.float32).pin_memory()
            n_h = 10
            for _ in range(2):
                m.predict_host_into(himg.numpy(), hdet.numpy(), hmask.numpy())
            t0 = time.perf_counter()
            for _ in range(n_h):
                m.predict_host_into(himg.numpy(), hdet.numpy(), hmask.numpy())
            dt_sync = (time.perf_counter() - t0) / n_h
            # ... and pipelined (mrcnn_maskrcnn_submit / _collect): the H2D of batch i + 1 crosses PCIe under the predict of batch i.
            # Two host image buffers alternate, as a host feeding a stream of batches would
            himg2 = himg.clone().pin_memory()
            bufs = (himg.numpy(), himg2.numpy())
            m.enable_timing(False)            # (the stage timer holds ONE batch's events: the library refuses two submissions in flight while it is on)
            m.submit(bufs[0])
            for i in range(2):
                m.submit(bufs[(i + 1) & 1]); m.collect(hdet.numpy(), hmask.numpy())
            t0 = time.perf_counter()
            for i in range(n_h):
                m.submit(bufs[(i + 1) & 1]); m.collect(hdet.numpy(), hmask.numpy())
            dt = (time.perf_counter() - t0) / n_h
            m.collect(hdet.numpy(), hmask.numpy())
            out["h2d_included"] = {"value": round(B / dt, 3), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": n_h,
                                   "synchronous_entry": {"value": round(B / dt_sync, 3), "ms_per_step": round(dt_sync * 1e3, 3)},
                                   "note": f"pinned host buffers in and out: {B * args.size * args.size * 3 / 1e6:.1f} MB H2D + "
                                           f"{(hdet.numel() + hmask.numel()) * 4 / 1e6:.1f} MB D2H per step inside the timing, through the pipelined "
                                           f"host entry (mrcnn_maskrcnn_submit / _collect: the copy of the next batch under the predict of this one); "
                                           f"synchronous_entry = mrcnn_maskrcnn_predict with host buffers, copy and compute back to back; never `value`"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], oracle_pred = cpu_baseline(model_dir, cfg, args, e2e_imgs)
            if n_e2e:
                out["parity_e2e"] = parity_e2e(e2e_pred, oracle_pred, n_e2e)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ---- the host-side group of an N > 1 run (gloo: rendezvous, barriers, the max-over-ranks of a float — never a GPU communicator) ----
# tests/test_host.py drives exactly these functions at world size 2 on CPU.
def host_group_init(rank, world):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:                     # --force-dist outside torchrun: a free port, not a fixed one
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if world > 1:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)


PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E spec
HBM_MEASURED_GBPS = 6290.0             # the streaming rate this engine's kernels have measured on the pool (profiles/r05_pmc_kernels_f32x3.txt: ROIAlign 6.28 TB/s)


def tile_class_entry(v, nbytes, ev_steps, ev_elapsed, peak_tflops):
    """One tile class of the conv family: its rate against BOTH roofs and which one bounds it.  ALGORITHMIC bytes (every operand across HBM once:
    conv_algorithmic_bytes in kernels_conv.hip) over the class's summed launch time; bound = whichever of flops / MFMA peak and bytes / 8 TB/s is larger."""
    launches, ms, flops = v
    tflops = flops / (ms * 1e-3) / 1e12
    gbps = nbytes / (ms * 1e-3) / 1e9
    t_mfma, t_hbm = flops / (peak_tflops * 1e12), nbytes / (PEAK_HBM_GBPS * 1e9)
    return {"launches_per_step": launches // max(ev_steps, 1), "share_of_step_time": round(ms / (1e3 * ev_elapsed), 4),
            "bound": "hbm" if t_hbm > t_mfma else "mfma",
            "tflops": round(tflops, 1), "frac_of_mfma": round(tflops / peak_tflops, 4),
            "algorithmic_bytes_per_launch": int(nbytes / launches), "gbps": round(gbps, 1),
            "frac_of_hbm": round(gbps / PEAK_HBM_GBPS, 4), "frac_of_hbm_measured": round(gbps / HBM_MEASURED_GBPS, 4)}


def shared_model_dir(rank, use_dist, write):
    """Rank 0 creates the model directory, writes it ONCE (write(dir)) and tells the others its path; they wait until it is complete."""
    import torch.distributed as dist
    box = [tempfile.mkdtemp(prefix="mrcnn_bench_") if rank == 0 else None]
    if use_dist:
        dist.broadcast_object_list(box, src=0)
    if rank == 0:
        write(box[0])
    if use_dist:
        dist.barrier()
    return box[0]


def broadcast_id(rank, make_id):
    """The 128-byte communicator id of the native exchange from rank 0 (make_id() -> bytes) to every rank."""
    import torch
    import torch.distributed as dist
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(make_id()), dtype=torch.uint8))
    dist.broadcast(idt, src=0)
    return bytes(idt.numpy().tobytes())


def gather_elapsed(elapsed, world):
    """Every rank's elapsed seconds, on every rank."""
    import torch
    import torch.distributed as dist
    every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(every, torch.tensor([elapsed], dtype=torch.float64))
    return [float(e.item()) for e in every]


def mfma_probe_live(seconds, dtype):
    """The live roof of this box (include/maskrcnn_hip_test.h: mrcnn_bench_mfma_probe): executed TFLOP/s of the matrix instruction the
    mode's convolutions issue, sustained over the last third of `seconds`."""
    import ctypes as C
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    tf, mhz = C.c_double(0), C.c_double(0)
    try:
        L.check(L.lib().mrcnn_bench_mfma_probe(float(seconds), L.F32 if dtype == "f32" else L.F16, C.byref(tf), C.byref(mhz)))
    except Exception as e:                       # the probe is a diagnostic: never fail the bench line over it
        sys.stderr.write(f"bench.py: live MFMA probe failed: {e}\n")
        return None
    return {"tflops": tf.value, "mhz": mhz.value, "seconds": seconds}


class SmiSampler:
    """Shader clock and socket power of one GPU sampled by a host thread through librocm_smi64 (the library behind rocm-smi: a call
    takes ~0.1 ms, the CLI half a second) while the timed loop runs.  Diagnostic only: every failure degrades to None."""

    def __init__(self, index):
        import ctypes as C
        self.C, self.index, self.lib, self.samples, self.thread, self.run = C, index, None, [], None, False
        for name in ("librocm_smi64.so", "librocm_smi64.so.7", "/opt/rocm/lib/librocm_smi64.so"):
            try:
                self.lib = C.CDLL(name)
                break
            except OSError:
                continue
        try:
            if self.lib is not None and self.lib.rsmi_init(C.c_uint64(0)) != 0:
                self.lib = None
        except Exception:
            self.lib = None

    def _once(self):
        C = self.C

        class Freq(C.Structure):
            _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]
        f = Freq()
        mhz = None
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.index), C.c_int(0), C.byref(f)) == 0 and f.current < 33:
            mhz = f.frequency[f.current] / 1e6
        w = None
        p, t = C.c_uint64(0), C.c_int(0)
        try:
            if self.lib.rsmi_dev_power_get(C.c_uint32(self.index), C.byref(p), C.byref(t)) == 0:
                w = p.value / 1e6
        except AttributeError:
            if self.lib.rsmi_dev_power_ave_get(C.c_uint32(self.index), C.c_uint32(0), C.byref(p)) == 0:
                w = p.value / 1e6
        return mhz, w

    def start(self):
        if self.lib is None:
            return
        import threading
        self.run = True

        def loop():
            while self.run:
                try:
                    self.samples.append(self._once())
                except Exception:
                    break
                time.sleep(0.02)
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()

    def stop(self):
        if self.lib is None or self.thread is None:
            return None
        self.run = False
        self.thread.join(timeout=2)
        clk = [c for c, _ in self.samples if c]
        pw = [w for _, w in self.samples if w]
        if not clk and not pw:
            return None
        return {"samples": len(self.samples), "sclk_mhz_mean": round(sum(clk) / len(clk), 0) if clk else None,
                "sclk_mhz_min": round(min(clk), 0) if clk else None, "sclk_mhz_max": round(max(clk), 0) if clk else None,
                "power_w_mean": round(sum(pw) / len(pw), 0) if pw else None,
                "how": "librocm_smi64 (rsmi_dev_gpu_clk_freq_get RSMI_CLK_TYPE_SYS / rsmi_dev_power_get) every ~20 ms from a host thread while the same step runs on for 0.5 s right behind the timed loop (untimed)"}


def sustained_peak(dtype, parts, achieved):
    """What the matrix cores of this board sustain for seconds with NO data movement (tools/probes/mfma_probe.hip under
    tools/mfma_power.sh, committed under profiles/): the fp16 MFMA on operands that change every instruction, the fp32 MFMA
    on constants.  Informational — `peak` / `frac` above stay the nominal figures of MI355X_MICROARCH.md."""
    path = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_mfma_power.txt") for r in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r02_mfma_power.txt"))
    try:
        want = "v_mfma_f32_32x32x2_f32" if dtype == "f32" else "random data"
        for line in open(path):
            if line.startswith(want) or want in line.split(":")[0]:
                tf = float(line.split(" s, ")[1].split(" TFLOP/s")[0])
                return {"value": round(tf / parts, 1), "unit": "TFLOP/s", "frac": round(achieved * parts / tf, 4),
                        "source": "profiles/" + os.path.basename(path) + " (whole chip, 6 s, no operand traffic; the board clocks down under matrix load)"}
    except (OSError, ValueError, IndexError):
        pass
    return None


CLASS_KERNELS = {"bneck": ("k_bneck_h<",), "128xNhalo": ("k_conv_halo<",), "128x256tail": ("k_conv_halo<",), "256x256pp": ("k_conv_pp<",), "128x128": (", 128, 1, 2, 4, 2,",), "128x64": (", 64, 1, 1, 4, 2,",),
                 "128x32": (", 32, 1, 1, 4, 1,",), "128x128w4": (", 128, 1, 4, 4, 1,",)}


def pmc_traffic(dtype, tile_class=None):
    """HBM bytes per launch of the dominant kernel CLASS (a tile class may be several template instantiations: the halo kernel
    with and without the fused head, 256- and 128-column tiles), launch-weighted, from the committed rocprofv3 PMC passes of THIS
    command in this compute mode (profiles/rNN_pmc_kernels_<dtype>.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE
    doubled per MI355X_MICROARCH.md §HBM).  PMC collection cannot run inside the timed process, hence a committed value."""
    if tile_class in CLASS_KERNELS:
        for rnd in ("r06", "r05", "r04", "r03"):
            try:
                with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_kernels_{dtype}.json")) as f:
                    ks = json.load(f)["kernels"]
                sel = [v for k, v in ks.items() if any(tag in k for tag in CLASS_KERNELS[tile_class]) and k.startswith(("k_conv", "k_bneck")) and "hbm_bytes_per_launch" in v]
                n = sum(v["launches_sampled"] for v in sel)
                if n:
                    return round(sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in sel) / n)
            except Exception:
                pass
    for name in (f"r06_pmc_traffic_{dtype}.json", f"r05_pmc_traffic_{dtype}.json", f"r04_pmc_traffic_{dtype}.json", f"r03_pmc_traffic_{dtype}.json", f"r02_pmc_traffic_{dtype}.json", "r01_pmc_traffic.json" if dtype == "f32" else None):
        if not name:
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return round(float(json.load(f)["hbm_bytes_per_launch_corrected"]))
        except Exception:
            continue
    return None


def pmc_traffic_source(dtype):
    for name in (f"r06_pmc_kernels_{dtype}.json", f"r05_pmc_kernels_{dtype}.json", f"r04_pmc_kernels_{dtype}.json", f"r03_pmc_kernels_{dtype}.json", f"r03_pmc_traffic_{dtype}.json", f"r02_pmc_traffic_{dtype}.json", "r01_pmc_traffic.json" if dtype == "f32" else None):
        if name and os.path.exists(os.path.join(ROOT, "profiles", name)):
            return "profiles/" + name
    return None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: run the N ranks of one node (one process per GPU, RCCL) and hand back
    rank 0's JSON line.  Fails loudly when the box has fewer than N GPUs."""
    try:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception as e:          # pragma: no cover
        sys.stderr.write(f"bench.py: cannot query the GPUs ({e})\n")
        return 3
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible on this box: not running (no n_gpus: 1 substitute)\n")
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def host_cores():
    """(physical, logical) core counts of the host."""
    logical = os.cpu_count() or 1
    physical = None
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        pass
    if not physical:
        try:
            seen = set()
            phys = core = None
            for line in open("/proc/cpuinfo"):
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
            physical = len(seen) or None
        except Exception:
            physical = None
    return int(physical or logical), int(logical)


def cpu_baseline(model_dir, cfg, args, imgs):
    """The oracle on host cores, to the reference's protocol (EvaluateCommand.swift:146-179): model load
    excluded, 1 warm-up image, then `cpu_images` images one at a time, wall clock around predict only; median
    ms/image → images/s.  Threads pinned to the physical core count.  The remaining images (up to --e2e-images)
    are predicted untimed: their outputs are the checker side of parity_e2e."""
    import numpy as np
    import torch
    from oracle.network import load_oracle_model
    physical, logical = host_cores()
    torch.set_num_threads(physical)
    os.environ["OMP_NUM_THREADS"] = str(physical)
    om = load_oracle_model(model_dir, cfg)
    dets, masks, ts = [], [], []
    n_total = max(1 + args.cpu_images, args.e2e_images if args.e2e_images > 0 else 0)
    for i in range(min(n_total, imgs.shape[0])):
        t0 = time.perf_counter()
        d, k = om.predict(imgs[i:i + 1])
        dt = time.perf_counter() - t0
        if 1 <= i <= args.cpu_images:
            ts.append(dt)
        dets.append(d[0]); masks.append(k[0])
    med = float(np.median(ts))
    rec = {"value": round(1.0 / med, 4), "unit": "images/s", "cores": int(torch.get_num_threads()), "physical_cores": physical,
           "logical_cores": logical, "kind": "port", "images": len(ts), "ms_per_image": {"median": round(med * 1e3, 1), "min": round(min(ts) * 1e3, 1), "max": round(max(ts) * 1e3, 1)},
           "note": "context only: the figure swings 0.14-0.24 images/s across boxes / rounds on 5 images — no GPU/CPU ratio should be quoted from it",
           "sample": f"{len(ts)} images of the same workload after 1 warm-up, batch 1, median {round(med * 1e3, 1)} ms/image "
                     f"(min {round(min(ts) * 1e3, 1)}, max {round(max(ts) * 1e3, 1)}); torch-CPU fp32 network (oneDNN, "
                     f"{int(torch.get_num_threads())} threads = physical cores) + C restatement of the custom layers (1 thread)"}
    return rec, (np.stack(dets), np.stack(masks))


def parity_e2e(hip_pred, oracle_pred, n):
    """HIP predict vs the CPU oracle's predict on the same n images, per compute mode: share of detections with the
    same class id and a box within 1e-4 (order-insensitive), and how far scores / masks of matched detections differ.
    Differences come from the convolutions' summation order (fp32 modes) or precision (f16) moving a score across a
    threshold / a neighbour in the NMS order; every stage is bit-exact on equal inputs (tests/)."""
    import numpy as np
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    od, ok = oracle_pred
    out = {"images": int(n), "box_tol": 1e-4, "modes": {}}
    for mode, (hd, hk) in hip_pred.items():
        tot_m = tot_d = same = 0
        identical = presence = 0
        ds = dm = dm_behind = 0.0
        n_behind = 0
        for i in range(n):
            a = ev.detection_agreement(hd[i], od[i], 1e-4, hk[i], ok[i])
            tot_m += a["matched"]; tot_d += max(a["n_a"], a["n_b"]); same += a["same_row"]
            identical += int(a["matched"] == max(a["n_a"], a["n_b"]))
            ds = max(ds, a["max_score_diff"]); dm = max(dm, a["max_mask_diff"]); presence += a["mask_presence_mismatch"]
            dm_behind = max(dm_behind, a["max_mask_diff_behind_presence_mismatch"]); n_behind += a["masks_behind_presence_mismatch"]
        out["modes"][mode] = {"detections": int(tot_d), "matched": int(tot_m), "fraction": round(tot_m / max(tot_d, 1), 4),
                              "same_rank": int(same), "images_fully_matched": int(identical),
                              "max_score_diff": float(np.float32(ds)), "max_mask_diff": float(np.float32(dm)),
                              "mask_presence_mismatch": int(presence), "masks_behind_presence_mismatch": int(n_behind),
                              "max_mask_diff_behind_presence_mismatch": float(np.float32(dm_behind))}
    return out


if __name__ == "__main__":
    main()
