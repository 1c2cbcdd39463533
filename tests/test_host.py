"""Host-side logic and the C-ABI surface, without a GPU: config / artefact tooling, the MLMultiArray
mirror, symbol export of libmaskrcnn_hip.so versus include/maskrcnn_hip.h, loud failure when no GPU
is present, and the N > 1 sharding + all-gather path on gloo (world_size 2 and 3)."""
import importlib
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_and_layer_params(pkg):
    cfg = pkg.ModelConfig()
    assert cfg.num_anchors() == 261888                       # A at 1024²
    assert pkg.ModelConfig(input_image_shape=(1536, 1536, 3)).num_anchors() == 589248
    p = cfg.proposal_layer_params()                          # keys of Conversion/task.py:25-35
    assert p == {"bboxStdDev_count": 4, "bboxStdDev_0": 0.1, "bboxStdDev_1": 0.1, "bboxStdDev_2": 0.2,
                 "bboxStdDev_3": 0.2, "preNMSMaxProposals": 6000, "maxProposals": 1000, "nmsIOUThreshold": 0.7}
    d = cfg.detection_layer_params()                         # task.py:57-67
    assert d["maxDetections"] == 100 and d["scoreThreshold"] == 0.7 and d["nmsIOUThreshold"] == 0.3
    assert cfg.pyramid_params(7) == {"poolSize": 7, "imageWidth": 1024, "imageHeight": 1024}
    c2 = pkg.ModelConfig.from_dict({"architecture": "resnet50", "input_image_shape": [512, 512, 3], "num_classes": 2,
                                    "pre_nms_max_proposals": 12000, "max_proposals": 500, "ignored_key": 1})
    assert (c2.architecture, c2.image_height, c2.num_classes, c2.pre_nms_max_proposals, c2.max_proposals) == \
        ("resnet50", 512, 2, 12000, 500)


def test_anchors_bin_layout(pkg, anchors_mod, tmp_path):
    cfg = pkg.ModelConfig(input_image_shape=(128, 128, 3))
    p = str(tmp_path / "anchors.bin")
    a = anchors_mod.write_anchors_bin(p, cfg)
    assert os.path.getsize(p) == a.shape[0] * 16             # A×4 float32 (task.py:176)
    b = anchors_mod.read_anchors_bin(p, cfg.num_anchors())
    np.testing.assert_array_equal(a, b)
    # first P2 anchor: centre (0,0), scale 32, ratio 0.5 → h = 32/sqrt(.5), w = 32*sqrt(.5); normalised by (127)
    h, w = 32 / np.sqrt(0.5), 32 * np.sqrt(0.5)
    want = (np.array([-h / 2, -w / 2, h / 2, w / 2]) - [0, 0, 1, 1]) / 127.0
    np.testing.assert_allclose(a[0], want.astype(np.float32), rtol=0, atol=1e-7)
    # level-major: the last anchors belong to P6 (scale 512)
    assert (a[-1, 2] - a[-1, 0]) > (a[0, 2] - a[0, 0]) * 8
    with open(p, "ab") as f:
        f.write(b"\0\0\0")
    with pytest.raises(ValueError):
        anchors_mod.read_anchors_bin(p)


def test_c_anchor_generator_matches_python(pkg, anchors_mod):
    """mrcnn_generate_anchors (host C, the reference's "generate the anchors on demand" TODO) is
    bit-identical to the Python generator that writes anchors.bin."""
    import ctypes as C
    lib_mod = importlib.import_module("mask-rcnn-coreml_amd._lib")
    if not os.path.exists(lib_mod.SO_PATH):
        pytest.skip("library not built")
    L = lib_mod.lib()
    for h, w in ((128, 128), (256, 192), (1024, 1024)):
        cfg = pkg.ModelConfig(input_image_shape=(h, w, 3))
        want = anchors_mod.generate_anchors(cfg)
        cnt = C.c_int64(0)
        lib_mod.check(L.mrcnn_generate_anchors(h, w, None, 0, C.byref(cnt)))
        assert cnt.value == want.shape[0] == cfg.num_anchors()
        got = np.empty((cnt.value, 4), np.float32)
        lib_mod.check(L.mrcnn_generate_anchors(h, w, got.ctypes.data, got.size, C.byref(cnt)))
        np.testing.assert_array_equal(got, want)
    with pytest.raises(lib_mod.MrcnnError):
        lib_mod.check(L.mrcnn_generate_anchors(64, 64, np.empty(8, np.float32).ctypes.data, 8, C.byref(cnt)))


def test_mrcw_roundtrip_and_synthetic_models(pkg, weights_mod, tmp_path):
    cfg = pkg.ModelConfig(architecture="resnet50", input_image_shape=(64, 64, 3), num_classes=5)
    paths = weights_mod.save_synthetic_models(str(tmp_path), cfg, seed=3)
    meta, t = weights_mod.read_mrcw(paths["MaskRCNN"])
    assert meta["kind"] == "MaskRCNN" and meta["architecture"] == "resnet50" and meta["num_classes"] == 5
    assert meta["ProposalLayer.preNMSMaxProposals"] == 6000 and meta["DetectionLayer.scoreThreshold"] == 0.7
    assert t["conv1/kernel"].shape == (64, 3, 7, 7) and t["conv1/kernel"].dtype == np.float16
    assert t["res4f_branch2b/kernel"].shape == (256, 256, 3, 3) and "res4g_branch2a/kernel" not in t
    assert t["rpn_class_raw/kernel"].shape == (6, 512, 1, 1) and t["rpn_bbox_pred/kernel"].shape == (12, 512, 1, 1)
    _, c = weights_mod.read_mrcw(paths["Classifier"])
    assert c["mrcnn_class_conv1/kernel"].shape == (1024, 256, 7, 7) and c["mrcnn_bbox_fc/kernel"].shape == (20, 1024)
    _, m = weights_mod.read_mrcw(paths["Mask"])
    assert m["mrcnn_mask_deconv/kernel"].shape == (256, 256, 2, 2) and m["mrcnn_mask/kernel"].shape == (5, 256, 1, 1)
    # determinism
    meta2, t2 = weights_mod.synthetic_models(cfg, seed=3)["MaskRCNN"]
    np.testing.assert_array_equal(t2["fpn_p2/kernel"], t["fpn_p2/kernel"])
    n101 = len(weights_mod.trunk_layers(pkg.ModelConfig(architecture="resnet101")))
    n50 = len(weights_mod.trunk_layers(cfg))
    assert n101 - n50 == 17 * 3


def test_mlmultiarray_shapes(pkg):
    ML = pkg.MLMultiArray
    a = ML(np.zeros((10, 4), np.float32))
    assert a.shape == (10, 1, 4, 1, 1) and a.strides == (4, 4, 1, 1, 1)
    b = ML(np.zeros((256, 8, 8), np.float32))
    assert b.shape == (1, 1, 256, 8, 8) and b.strides[2] == 64
    c = ML(np.zeros((3, 1, 256, 7, 7), np.float32))
    assert c.strides[0] == 256 * 49
    with pytest.raises(TypeError):
        ML(np.zeros((4, 4), np.float64))


def test_c_abi_exports_every_declared_symbol(pkg):
    lib_mod = importlib.import_module("mask-rcnn-coreml_amd._lib")
    if not os.path.exists(lib_mod.SO_PATH):
        pytest.skip("libmaskrcnn_hip.so not built (run python __graft_entry__.py)")
    L = lib_mod.lib()
    for header, names, floor in (("maskrcnn_hip.h", lib_mod.EXPORTED_SYMBOLS, 25), ("maskrcnn_hip_test.h", lib_mod.TEST_SYMBOLS, 5)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = sorted(set(re.findall(r"MRCNN_API\s+[\w\s\*]+?\b(mrcnn_\w+)\s*\(", hdr)))
        assert len(declared) >= floor
        missing = [s for s in declared if not hasattr(L, s)]
        assert not missing, (header, missing)
        assert sorted(names) == declared, header
    # the drop-in header carries no test / measurement knob
    prod = open(os.path.join(ROOT, "include", "maskrcnn_hip.h")).read()
    for knob in ("mrcnn_debug_set", "mrcnn_bench_conv", "mrcnn_conv2d_nhwc", "mrcnn_model_conv_profile"):
        assert knob + "(" not in prod.replace(" (", "("), knob
    assert L.mrcnn_version().startswith(b"maskrcnn_hip")


def test_no_cpu_fallback_without_gpu(pkg, tmp_path):
    """On a box without a GPU every compute entry point fails loudly with MRCNN_ERR_HIP."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib_mod = importlib.import_module("mask-rcnn-coreml_amd._lib")
    if not os.path.exists(lib_mod.SO_PATH):
        pytest.skip("library not built")
    with pytest.raises(lib_mod.MrcnnError) as e:
        pkg.DetectionLayer({})
    assert e.value.code == 3 and "no CPU fallback" in str(e.value)
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    with pytest.raises(lib_mod.MrcnnError) as e:
        models.Classifier(str(tmp_path / "x.mrcw"))
    assert e.value.code == 3
    # host-only entry points work without a GPU
    det = np.array([[0.1, 0.2, 0.5, 0.8, 3, 0.9], [0, 0, 0, 0, 0, 0]], np.float32)
    d = pkg.Detection.detectionsFromFeatureValue(det, np.full((2, 28, 28), 0.5, np.float32))
    assert len(d) == 1 and d[0].classId == 3 and d[0].mask[0, 0] == 191
    assert abs(pkg.IOU((0, 0, 1, 1), (0.5, 0, 1, 1)) - 1 / 3) < 1e-6


def _results_proto_classes():
    """The reference's results.proto rebuilt with google.protobuf from the field numbers in
    Sources/maskrcnn/results.pb.swift — an independent codec to check the hand-written one against."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="results.proto", package="maskrcnn", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for no, fname, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=no, type=ftype, label=label)
            if tname:
                f.type_name = ".maskrcnn." + tname
    O, R = T.LABEL_OPTIONAL, T.LABEL_REPEATED
    msg("Origin", [(1, "x", T.TYPE_DOUBLE, O, None), (2, "y", T.TYPE_DOUBLE, O, None)])
    msg("Size", [(1, "width", T.TYPE_DOUBLE, O, None), (2, "height", T.TYPE_DOUBLE, O, None)])
    msg("Rect", [(1, "origin", T.TYPE_MESSAGE, O, "Origin"), (2, "size", T.TYPE_MESSAGE, O, "Size")])
    msg("ImageInfo", [(1, "datasetId", T.TYPE_STRING, O, None), (2, "id", T.TYPE_STRING, O, None),
                      (3, "width", T.TYPE_INT32, O, None), (4, "height", T.TYPE_INT32, O, None)])
    msg("Detection", [(1, "probability", T.TYPE_DOUBLE, O, None), (2, "classId", T.TYPE_INT32, O, None),
                      (3, "classLabel", T.TYPE_STRING, O, None), (4, "boundingBox", T.TYPE_MESSAGE, O, "Rect")])
    msg("Result", [(1, "imageInfo", T.TYPE_MESSAGE, O, "ImageInfo"), (2, "detections", T.TYPE_MESSAGE, R, "Detection")])
    msg("Results", [(1, "results", T.TYPE_MESSAGE, R, "Result")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("maskrcnn.Results"))


def test_results_proto_wire_format():
    rp = importlib.import_module("mask-rcnn-coreml_amd.results_pb")
    det = np.zeros((5, 6), np.float32)
    det[0] = [0.1, 0.2, 0.5, 0.8, 17, 0.9]
    det[1] = [0.0, 0.0, 1.0, 1.0, 3, 0.7]            # not > 0.7: dropped (EvaluateCommand.swift:216)
    det[2] = [0.25, 0.5, 0.75, 1.0, 1, 1.0]
    pbs = rp.detections_to_pb(det)
    assert [d.classId for d in pbs] == [17, 1] and all(d.classLabel == "test" for d in pbs)
    assert pbs[0].x == float(np.float32(0.2)) and pbs[0].height == float(np.float32(0.5)) - float(np.float32(0.1))
    results = [rp.PBResult("coco", "139", 640, 426, pbs), rp.PBResult("coco", "285", 0, 0, [])]
    data = rp.encode_results(results)
    assert rp.decode_results(data) == results                                # own round trip
    Results = _results_proto_classes()
    m = Results()
    m.ParseFromString(data)                                                  # protobuf accepts our bytes
    assert len(m.results) == 2 and m.results[0].imageInfo.id == "139" and m.results[0].imageInfo.width == 640
    d0 = m.results[0].detections[0]
    assert (d0.probability, d0.classId, d0.classLabel) == (pbs[0].probability, 17, "test")
    assert (d0.boundingBox.origin.x, d0.boundingBox.size.height) == (pbs[0].x, pbs[0].height)
    assert m.SerializeToString(deterministic=True) == data                   # and produces the same bytes


def test_oracle_and_product_do_not_mix():
    """The product package never imports the oracle."""
    pkg_dir = os.path.join(ROOT, "mask-rcnn-coreml_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "from oracle" not in txt and "import oracle" not in txt and "mrcnn_oracle" not in txt, f


# ---------------------------------------------------------------------------------------------------
# multi-rank path on gloo
# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_predict(images):
    """Deterministic per-image function standing in for the GPU model (independent of the batch)."""
    import torch
    x = torch.as_tensor(images).to(torch.float32)
    s = x.reshape(x.shape[0], -1).sum(1)
    det = torch.stack([s + k for k in range(4 * 6)], 1).reshape(-1, 4, 6)
    mask = torch.stack([s * 0.5 + k for k in range(4 * 9)], 1).reshape(-1, 4, 3, 3)
    return det, mask


def _rank_main(rank, world, port, global_batch, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    imgs = torch.arange(global_batch * 2 * 2 * 3, dtype=torch.float32).reshape(global_batch, 2, 2, 3)
    det, mask = dmod.predict_sharded(_fake_predict, imgs, 4, 3)
    # equal-shard pre-allocated path (the bench's)
    lo, hi = dmod.shard_bounds(world * 2, world, rank)
    g = dmod.DetectionGather(2, 4, 3, world, "cpu")
    d2, m2 = _fake_predict(imgs[:world * 2][lo:hi])
    gd, gm = g.all_gather(d2, m2)
    q.put((rank, det.numpy(), mask.numpy(), gd.numpy().copy(), gm.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,global_batch", [(2, 8), (3, 7)])
def test_sharded_predict_gloo(world, global_batch):
    import torch
    import torch.multiprocessing as mp
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    bounds = [dmod.shard_bounds(global_batch, world, r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == global_batch
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    imgs = torch.arange(global_batch * 2 * 2 * 3, dtype=torch.float32).reshape(global_batch, 2, 2, 3)
    want_d, want_m = _fake_predict(imgs)
    for rank, det, mask, gd, gm in res:
        np.testing.assert_array_equal(det, want_d.numpy())          # every rank holds the whole batch, in order
        np.testing.assert_array_equal(mask, want_m.numpy())
        wd, wm = _fake_predict(imgs[:world * 2])
        np.testing.assert_array_equal(gd, wd.numpy())
        np.testing.assert_array_equal(gm, wm.numpy())


def _rank_main_failing(rank, world, port, global_batch, mode, q):
    """mode "status": rank 1's local predict raises — it still joins the exchange with a zeroed slot and its status word.
    mode "gone": rank 1 cannot reach the collective at all and tears its end down (the gloo stand-in for release_peers' ncclCommAbort)."""
    import datetime
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    imgs = torch.arange(global_batch * 2 * 2 * 3, dtype=torch.float32).reshape(global_batch, 2, 2, 3)

    class WatchdogError(RuntimeError):
        code = 5                                   # MRCNN_ERR_UNSUPPORTED: the fp16-range watchdog

    def predict(x):
        if rank == 1:
            raise WatchdogError("an activation left the fp16 range")
        return _fake_predict(x)

    def report(*rec):
        q.put(rec)
        q.close()
        q.join_thread()                            # (flushed before the process goes away without the interpreter's own shutdown)

    t0 = time.monotonic()
    try:
        if mode == "gone" and rank == 1:
            report(rank, "gone", "", 0.0)
            os._exit(0)                            # never reaches the collective: its sockets close, as an aborted communicator's would
        dmod.predict_sharded(predict, imgs, 4, 3)
        report(rank, "returned", "", time.monotonic() - t0)
    except dmod.RemoteRankError as e:
        report(rank, "remote", f"{e.statuses}|{e}|{type(e.__cause__).__name__}", time.monotonic() - t0)
    except Exception as e:                         # the peers of a rank that is gone: their collective FAILS instead of blocking
        report(rank, "failed", f"{type(e).__name__}: {e}"[:300], time.monotonic() - t0)
    os._exit(0)


@pytest.mark.parametrize("mode", ["status", "gone"])
def test_a_rank_that_fails_before_the_collective_does_not_hang_its_peers(mode):
    """world 2 over gloo, rank 1 fails BEFORE the all-gather.
    "status": its local predict raises — the twin of csrc/dist.hip's issue_exchange sends a zeroed slot with the status word, and BOTH
    ranks raise RemoteRankError naming rank 1 (rank 1's chained to its own exception).  "gone": rank 1 cannot take part at all and tears
    its end down (release_peers: ncclCommAbort) — rank 0's collective fails within the group's timeout instead of blocking for ever."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main_failing, args=(r, 2, port, 5, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, kind, text, secs = q.get(timeout=90)
        res[rank] = (kind, text, secs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "status":
        assert res[0][0] == "remote" and res[1][0] == "remote", res
        assert res[0][1].startswith("[0, 5]|") and "rank(s) [1] of 2 failed" in res[0][1] and res[0][1].endswith("|NoneType")
        assert res[1][1].startswith("[0, 5]|") and "see this rank's earlier message" in res[1][1] and res[1][1].endswith("|WatchdogError")
        assert res[0][2] < 15 and res[1][2] < 15
    else:
        assert res[1][0] == "gone"
        assert res[0][0] == "failed", res           # an error, not a hang (and not a silently short result)
        assert res[0][2] < 45, res


def test_native_dist_aborted_rank_fails_the_exchange_on_its_peers():
    """csrc/dist.hip release_peers: a rank that cannot even enqueue a zeroed slot aborts the communicator, so that its peers'
    ncclAllGather fails (nccl_check raises) instead of blocking.  Through the host seam: status MRCNN_DIST_ABORTED for rank 1 makes
    the exchange fail with MRCNN_ERR_HIP naming the rank, and no output word is written."""
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    D, S, world, gb = 3, 2, 2, 5
    rng = np.random.default_rng(6)
    det = rng.standard_normal((gb, D, 6)).astype(np.float32)
    mask = rng.standard_normal((gb, D, S, S)).astype(np.float32)
    bounds = [dmod.shard_bounds(gb, world, r) for r in range(world)]
    with pytest.raises(L.MrcnnError) as ei:
        dmod.NativeDist.simulate_host([det[b:e] for b, e in bounds], [mask[b:e] for b, e in bounds], gb, D, S, np.array([0, -1], np.int32))
    assert ei.value.code == 3 and "rank 1 aborted the communicator" in str(ei.value)          # MRCNN_ERR_HIP


def test_the_knobs_are_out_of_a_production_hosts_reach():
    """VERDICT r5 weak 9 / ADVICE r5: the alternative kernel forms behind mrcnn_debug_set and the MRCNN_* overrides of their defaults are
    test / measurement equipment.  In a process that was NOT started with MRCNN_TEST_KNOBS=1 (every production host) mrcnn_debug_set
    answers MRCNN_ERR_UNSUPPORTED, names the switch in its message and changes nothing; with it (this test process: tests/conftest.py)
    the same call is accepted.  No GPU needed: the check sits in front of everything else."""
    import subprocess
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    L.check(L.lib().mrcnn_debug_set(b"conv_bneck", 1))                       # armed here
    assert L.lib().mrcnn_debug_set(b"no_such_knob", 1) != 0
    code = ("import importlib, sys; sys.path.insert(0, %r); L = importlib.import_module('mask-rcnn-coreml_amd._lib'); "
            "rc = L.lib().mrcnn_debug_set(b'conv_bneck', 0); print(rc, L.lib().mrcnn_last_error().decode())" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "MRCNN_TEST_KNOBS"}
    env["MRCNN_BNECK"] = "0"                                                  # a stray override in a production environment: ignored
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == 5 and "MRCNN_TEST_KNOBS=1" in msg and "conv_bneck" in msg        # MRCNN_ERR_UNSUPPORTED


def test_unletterbox_boxes_follow_the_norm_boxes_convention(pkg):
    """normalized coordinates are Matterport's norm_boxes (pixel = n*(size-1), far edge +1) — the convention
    of anchors.py and the mask paste — so a box covering exactly the letterboxed content maps back to the full
    source frame, and the content's centre pixel maps to the source's centre."""
    ev = __import__("importlib").import_module("mask-rcnn-coreml_amd.evaluate")
    h, w, H, W = 480, 640, 1024, 1024
    nh, nw, py, px = ev.letterbox_geometry(h, w, H, W)
    assert (nh, nw, py, px) == (768, 1024, 128, 0)
    content = np.array([[py / (H - 1), px / (W - 1), (py + nh - 1) / (H - 1), (px + nw - 1) / (W - 1), 1, 0.9]])
    back = ev.unletterbox_boxes(content, h, w, H, W)
    np.testing.assert_allclose(back[0, :4], [0.0, 0.0, 1.0, 1.0], atol=1e-12)
    # a 2:1 down-scaled interior box: pixel rows [228, 428) of the canvas = source rows [62.5, 187.5)
    box = np.array([[228 / (H - 1), 100 / (W - 1), (428 - 1) / (H - 1), (300 - 1) / (W - 1), 1, 0.9]])
    back = ev.unletterbox_boxes(box, h, w, H, W)
    sy, sx = h / nh, w / nw
    np.testing.assert_allclose(back[0, 0] * (h - 1), (228 - py) * sy, atol=1e-9)
    np.testing.assert_allclose(back[0, 2] * (h - 1) + 1, (428 - py) * sy, atol=1e-9)
    np.testing.assert_allclose(back[0, 1] * (w - 1), 100 * sx, atol=1e-9)
    np.testing.assert_allclose(back[0, 3] * (w - 1) + 1, 300 * sx, atol=1e-9)
    # boxes in the black borders clip to the frame
    border = np.array([[0.0, 0.0, 50 / (H - 1), 1.0, 1, 0.9]])
    assert ev.unletterbox_boxes(border, h, w, H, W)[0, 2] == 0.0
    # zero-padded rows (the detections array always has maxDetections rows) stay all-zero, in the mirror and in the C entry:
    # for a down-scaled source an all-zero box would otherwise come back with a non-zero far edge
    padded = np.zeros((4, 6), np.float32)
    padded[0] = [0.3, 0.3, 0.6, 0.6, 2, 0.9]
    back = ev.unletterbox_boxes(padded, 2000, 3000, 1024, 1024)
    assert np.all(back[1:] == 0.0) and np.any(back[0, :4] != padded[0, :4])
    lib_mod = __import__("importlib").import_module("mask-rcnn-coreml_amd._lib")
    if os.path.exists(lib_mod.SO_PATH):
        c = padded.copy()
        lib_mod.check(lib_mod.lib().mrcnn_unletterbox_boxes(c.ctypes.data, 4, 6, 2000, 3000, 1024, 1024))
        assert np.all(c[1:] == 0.0)
        np.testing.assert_allclose(c[0, :4], back[0, :4].astype(np.float32), atol=1e-7)


def test_detection_agreement_counts(pkg):
    ev = __import__("importlib").import_module("mask-rcnn-coreml_amd.evaluate")
    a = np.zeros((6, 6), np.float32)
    a[0] = [0.1, 0.1, 0.5, 0.5, 3, 0.99]
    a[1] = [0.2, 0.2, 0.6, 0.6, 7, 0.95]
    a[2] = [0.3, 0.3, 0.7, 0.7, 7, 0.90]
    b = a.copy()
    r = ev.detection_agreement(a, b)
    assert r["matched"] == 3 and r["fraction"] == 1.0 and r["same_row"] == 3
    b[[1, 2]] = b[[2, 1]]                         # neighbours swapped: still the same set
    r = ev.detection_agreement(a, b)
    assert r["matched"] == 3 and r["same_row"] == 1
    b[0, 4] = 4                                   # class differs
    b[2, 0] += 5e-4                               # box moved beyond the tolerance
    r = ev.detection_agreement(a, b)
    assert r["matched"] == 1 and abs(r["fraction"] - 1 / 3) < 1e-12
    b[3] = [0.4, 0.4, 0.8, 0.8, 1, 0.8]           # an extra detection on one side lowers the fraction
    assert ev.detection_agreement(a, b)["fraction"] == 0.25
    assert ev.detection_agreement(np.zeros((4, 6)), np.zeros((4, 6)))["fraction"] == 1.0


def test_detection_agreement_separates_masks_behind_a_dropped_row(pkg):
    """A mask emptied by removeZeros on one side only is counted, and the pairs behind it (whose class lookup the reference's
    compact index shifts, TimeDistributedMaskLayer.swift:71) do not enter `max_mask_diff`."""
    ev = __import__("importlib").import_module("mask-rcnn-coreml_amd.evaluate")
    a = np.zeros((5, 6), np.float32)
    for i in range(4):
        a[i] = [0.1 * i, 0.1 * i, 0.1 * i + 0.3, 0.1 * i + 0.3, 1 + i, 0.99 - 0.01 * i]
    ma = np.full((5, 4, 4), 0.5, np.float32)
    ma[4] = 0
    mb = ma.copy()
    mb[0] += 1e-5
    r = ev.detection_agreement(a, a.copy(), 1e-4, ma, mb)
    assert r["mask_presence_mismatch"] == 0 and abs(r["max_mask_diff"] - 1e-5) < 1e-7 and r["masks_behind_presence_mismatch"] == 0
    mb[1] = 0                                      # dropped on side B only
    mb[2] = 0.9                                    # ... so the rows behind it carry another class's mask
    r = ev.detection_agreement(a, a.copy(), 1e-4, ma, mb)
    assert r["mask_presence_mismatch"] == 1 and r["matched"] == 4
    assert abs(r["max_mask_diff"] - 1e-5) < 1e-7
    assert r["masks_behind_presence_mismatch"] == 2 and abs(r["max_mask_diff_behind_presence_mismatch"] - 0.4) < 1e-6


def test_bench_sustained_peak_reads_the_committed_probe():
    """bench.py's `roofline.sustained_peak` comes from the committed probe output (profiles/r02_mfma_power.txt): the fp16 MFMA
    rate on changing operands divided by the passes a product takes; the fp32 MFMA rate for the fp32 mode."""
    import bench
    sp = bench.sustained_peak("f32x3", 3, 300.0)
    assert sp is not None and 400 < sp["value"] < 700 and abs(sp["frac"] - 300.0 / sp["value"]) < 1e-3
    assert abs(bench.sustained_peak("f16", 1, 500.0)["value"] - 3 * sp["value"]) < 1.0
    assert 140 < bench.sustained_peak("f32", 1, 100.0)["value"] < 160


def test_bench_tile_class_entry_names_the_roof_that_bounds_a_class():
    """bench.py's per-class record (round 6): a class is hbm-bound when its ALGORITHMIC bytes at 8 TB/s take longer than its flops at the MFMA peak;
    both rates and both fractions are carried either way."""
    import bench
    # 62 launches, 4.67 ms, 1.115e12 flop and 1.83e10 B in total: the 1x1 class of the headline mode — 239 TFLOP/s, 3.9 TB/s
    e = bench.tile_class_entry((62, 4.67, 1.115e12), 1.83e10, 1, 18.0e-3, 2500.0 / 3)
    assert e["bound"] == "hbm" and e["launches_per_step"] == 62
    assert abs(e["tflops"] - 238.8) < 0.5 and abs(e["gbps"] - 3918.6) < 2 and abs(e["frac_of_hbm"] - 0.4898) < 1e-3 and abs(e["frac_of_hbm_measured"] - 0.623) < 1e-3
    assert abs(e["share_of_step_time"] - 4.67 / 18.0) < 1e-4 and e["algorithmic_bytes_per_launch"] == int(1.83e10 / 62)
    # the 3x3 class: 46 launches, 9.46 ms, 4.48e12 flop, 6.5e9 B — far on the matrix side
    e = bench.tile_class_entry((46, 9.46, 4.48e12), 6.5e9, 1, 18.0e-3, 2500.0 / 3)
    assert e["bound"] == "mfma" and abs(e["frac_of_mfma"] - 0.5683) < 1e-3 and e["frac_of_hbm"] < 0.1


def test_bench_host_core_count():
    import bench
    physical, logical = bench.host_cores()
    assert 1 <= physical <= logical == os.cpu_count()


def test_native_dist_shard_matches_the_torch_twin(pkg):
    """mrcnn_dist_shard (C ABI, host arithmetic) = dist.shard_bounds for every (batch, world, rank): contiguous blocks that
    differ by at most one image and cover the batch exactly; record geometry = 316 000 B at the defaults."""
    import ctypes as C
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    for world in (1, 2, 3, 4, 7, 8):
        for batch in (0, 1, 5, 8, 31, 32, 64, 65):
            covered = []
            for rank in range(world):
                lo, hi = dmod.NativeDist.shard(batch, world, rank)
                assert (lo, hi) == dmod.shard_bounds(batch, world, rank)
                covered += list(range(lo, hi))
            assert covered == list(range(batch))
    assert L.lib().mrcnn_dist_record_floats(100, 28) * 4 == 316000
    lo, hi = C.c_int(), C.c_int()
    assert L.lib().mrcnn_dist_shard(8, 4, 4, C.byref(lo), C.byref(hi)) != 0          # rank out of range: status, no abort
    assert b"dist_shard" in L.lib().mrcnn_last_error()


def test_coco_instances_reader(pkg, tmp_path):
    """COCO.swift counterpart: decode instances_*.json, index annotations by image id in file order, iterate sorted by id with a
    limit (EvaluateCommand.swift:165 uses limit 5, sortById true); a limit beyond the image count is an error, like the Swift slice."""
    import json
    coco_mod = importlib.import_module("mask-rcnn-coreml_amd.coco")
    doc = {"info": {"description": "t", "url": "http://x", "version": "1", "year": 2017, "contributor": "c"},
           "images": [{"id": 42, "file_name": "b.jpg", "width": 640, "height": 480, "coco_url": "ignored"},
                      {"id": 7, "file_name": "a.jpg", "width": 500, "height": 375},
                      {"id": 19, "file_name": "c.jpg", "width": 64, "height": 64}],
           "annotations": [{"id": 1, "image_id": 42, "category_id": 18, "bbox": [1.5, 2, 30, 40], "iscrowd": 0},
                           {"id": 2, "image_id": 7, "category_id": 1, "bbox": [0, 0, 10, 10]},
                           {"id": 3, "image_id": 42, "category_id": 3, "bbox": [5, 5, 6, 7]}]}
    p = tmp_path / "instances_val.json"
    p.write_text(json.dumps(doc))
    coco = pkg.COCO(str(p))
    assert [i.id for i in coco.images] == [42, 7, 19]
    assert [a.id for a in coco.index[42]] == [1, 3] and coco.index[42][0].bbox == (1.5, 2.0, 30.0, 40.0)
    got = list(coco.makeImageIterator(limit=2, sortById=True))
    assert [(im.id, im.fileName, im.width, im.height, [a.id for a in anns]) for im, anns in got] == [
        (7, "a.jpg", 500, 375, [2]), (19, "c.jpg", 64, 64, [])]
    assert [im.id for im, _ in coco.makeImageIterator()] == [42, 7, 19]                  # file order without sortById
    with pytest.raises(ValueError):
        list(coco.makeImageIterator(limit=4))
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"images": [], "annotations": []}))
    with pytest.raises(ValueError, match="info"):
        coco_mod.COCO(str(bad))


def test_mask_to_u8_double_path_equals_float_path():
    """Detection.swift:77-98 quantises a Double mask; the fp32 mask widened to double must give the same bytes as the float entry."""
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    m = np.random.default_rng(4).random(784 * 3).astype(np.float32)
    m[:4] = [0.0, 1.0, 0.5, 2.0]
    a, b = np.empty(m.size, np.uint8), np.empty(m.size, np.uint8)
    L.check(L.lib().mrcnn_mask_to_u8(m.ctypes.data, m.size, a.ctypes.data))
    md = m.astype(np.float64)
    L.check(L.lib().mrcnn_mask_to_u8_f64(md.ctypes.data, m.size, b.ctypes.data))
    np.testing.assert_array_equal(a, b)
    assert a[0] == 255 and a[1] == 127 and a[3] == 0


# ---------------------------------------------------------------------------------------------------
# the native exchange (csrc/dist.hip) through its host seam, and bench.py's launcher
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_native_dist_layout_matches_the_torch_twin(world):
    """csrc/dist.hip's pack -> all-gather -> unpack (the code the GPU path runs, copy primitive swapped for memcpy) at world
    sizes 1-8 with even and uneven shards, against mask-rcnn-coreml_amd/dist.py's record layout and shard arithmetic —
    the native path cannot run at world > 1 on the build machine."""
    import torch
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    D, S = 5, 4
    rng = np.random.default_rng(world)
    for global_batch in sorted({1, world, world + 1, 2 * world - 1 if world > 1 else 3, 8 * world, 8 * world + 3}):
        det = rng.standard_normal((global_batch, D, 6)).astype(np.float32)
        mask = rng.standard_normal((global_batch, D, S, S)).astype(np.float32)
        plan, slot = dmod.NativeDist.plan(global_batch, world, D, S)
        n_max = -(-global_batch // world)
        rec = D * 6 + D * S * S
        assert slot == n_max * rec + 4
        bounds = [dmod.shard_bounds(global_batch, world, r) for r in range(world)]
        assert [(b, e) for b, e, _, _ in plan] == bounds
        assert [o for _, _, o, _ in plan] == [r * slot for r in range(world)]
        assert [c for _, _, _, c in plan] == [(e - b) * rec for b, e in bounds]
        for r in range(world):
            assert dmod.NativeDist.shard(global_batch, world, r) == bounds[r]
        locd = [det[b:e] for b, e in bounds]
        locm = [mask[b:e] for b, e in bounds]
        od, om, st = dmod.NativeDist.simulate_host(locd, locm, global_batch, D, S)
        np.testing.assert_array_equal(od, det)
        np.testing.assert_array_equal(om, mask)
        assert not st.any()
        # the torch twin's record layout: pack -> pad to the largest shard -> concatenate -> drop padding -> unpack
        parts = []
        for (b, e), d_, m_ in zip(bounds, locd, locm):
            pad = torch.zeros((n_max, rec))
            if e > b:
                pad[: e - b] = dmod.pack_records(torch.from_numpy(d_), torch.from_numpy(m_))
            parts.append(pad[: e - b])
        td, tm = dmod.unpack_records(torch.cat(parts, 0), D, S)
        np.testing.assert_array_equal(od, td.numpy())
        np.testing.assert_array_equal(om, tm.numpy())


def test_native_dist_failed_rank_still_contributes_a_slot():
    """a rank whose local predict failed takes part in the collective with zeroed records and its status
    word; every rank sees the status (and raises) instead of blocking in ncclAllGather."""
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    D, S, world, gb = 3, 2, 4, 9
    rng = np.random.default_rng(5)
    det = rng.standard_normal((gb, D, 6)).astype(np.float32)
    mask = rng.standard_normal((gb, D, S, S)).astype(np.float32)
    bounds = [dmod.shard_bounds(gb, world, r) for r in range(world)]
    status = np.array([0, 0, 5, 0], np.int32)        # rank 2: MRCNN_ERR_UNSUPPORTED (the fp16-range watchdog)
    od, om, st = dmod.NativeDist.simulate_host([det[b:e] for b, e in bounds], [mask[b:e] for b, e in bounds], gb, D, S, status)
    np.testing.assert_array_equal(st, status)
    b2, e2 = bounds[2]
    assert not od[b2:e2].any() and not om[b2:e2].any()          # zeroed, not garbage
    keep = np.r_[0:b2, e2:gb]
    np.testing.assert_array_equal(od[keep], det[keep])
    np.testing.assert_array_equal(om[keep], mask[keep])


def test_bench_gpus_n_fails_loudly_without_n_gpus():
    """`python bench.py --gpus 2` is the driver's command line for N = 2: without two GPUs it must fail, not print an
    n_gpus: 1 line; a --gpus / WORLD_SIZE mismatch is an error too."""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not r.stdout.strip(), r.stdout                       # no JSON line
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and not r.stdout.strip() and "WORLD_SIZE" in r.stderr


def test_native_dist_binds_the_rccl_copy_the_process_already_holds(pkg):
    """VERDICT r3 item 8 — one RCCL per process: dist.hip takes the copy the host has already mapped (PyTorch ships one under the
    soname librccl.so.1) before it loads its own.  Two fresh interpreters: with torch imported first the library reports the
    shared copy (1), without torch its own dlopen (0); MRCNN_RCCL_PRIVATE=1 forces the private load.  No GPU needed."""
    import subprocess
    import sys
    lib_mod = importlib.import_module("mask-rcnn-coreml_amd._lib")
    if not os.path.exists(lib_mod.SO_PATH):
        pytest.skip("libmaskrcnn_hip.so not built")
    probe = ("import ctypes, sys; {pre}; L = ctypes.CDLL({so!r}); L.mrcnn_dist_rccl_shared.restype = ctypes.c_int; "
             "print('shared', L.mrcnn_dist_rccl_shared())")
    def run(pre, env=None):
        r = subprocess.run([sys.executable, "-c", probe.format(pre=pre, so=lib_mod.SO_PATH)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stderr
        return int(r.stdout.strip().split()[-1])
    assert run("import torch") == 1
    assert run("pass") == 0
    assert run("import torch", {"MRCNN_RCCL_PRIVATE": "1"}) == 0


def _bench_host_group_rank(rank, world, port, q):
    """One rank of bench.py's host-side group, environment as torchrun sets it."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    bench = importlib.import_module("bench")
    bench.host_group_init(rank, world)
    wrote = []
    d = bench.shared_model_dir(rank, True, lambda path: (open(os.path.join(path, "model.bin"), "wb").write(b"x" * 1000), wrote.append(path)))
    ident = bench.broadcast_id(rank, (lambda: bytes(range(128))) if rank == 0 else None)
    every = bench.gather_elapsed(1.0 + rank, world)
    q.put((rank, d, os.path.getsize(os.path.join(d, "model.bin")), len(wrote), ident, every))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_host_group_at_world_two_on_gloo():
    """the N > 1 control flow of bench.py that never ran anywhere — its torch process group is gloo (host-side
    only: ONE RCCL user per process, the native exchange), rank 0 writes the model directory ONCE and every rank reads the complete
    files, the 128-byte communicator id reaches every rank, every rank learns every rank's time (the job is as slow as its slowest
    rank).  The same functions bench.py calls, two processes, no GPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_host_group_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, d0, sz0, w0, id0, e0), (r1, d1, sz1, w1, id1, e1) = res
    assert d0 == d1 and sz0 == sz1 == 1000 and (w0, w1) == (1, 0)          # one writer, both see the finished file
    assert id0 == id1 == bytes(range(128))
    assert e0 == e1 == [1.0, 2.0]
