#!/opt/conda/bin/python3.9
"""Generates tests/golden/keras_tiny.h5.gz + keras_tiny.npz with the REAL h5py/libhdf5 (the image's
/opt/conda/bin/python3.9 has h5py 3.3 / HDF5 1.10.6; the system interpreter has none).

The file is laid out the way ``keras.engine.topology.save_weights_to_hdf5_group`` (Keras 2.1.6, the
version the reference's converter pins, Conversion/requirements.txt) lays out a Matterport Mask R-CNN
checkpoint: root attrs ``layer_names`` / ``backend`` / ``keras_version``; one group per layer with a
``weight_names`` attr; datasets at ``/<layer>/<weight_name>`` where weight names contain a slash
(``conv1/kernel:0``) so they sit one group deeper; a nested model (``rpn_model``) holds several
layers' weights; layers without weights are empty groups.  Enough groups (300) that the root group's
B-tree has two levels, like the real 390-layer checkpoint.

It pins mask-rcnn-coreml_amd/hdf5.py (an independent pure-Python parser) against the reference HDF5
implementation:  /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py
"""
import gzip
import os
import shutil

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(1234)


def f32(*shape):
    return rng.standard_normal(shape).astype(np.float32)


layers = []          # (layer_name, [(weight_name, array)])
layers.append(("input_image", []))
layers.append(("conv1", [("conv1/kernel:0", f32(7, 7, 3, 8)), ("conv1/bias:0", f32(8))]))
layers.append(("bn_conv1", [("bn_conv1/gamma:0", f32(8)), ("bn_conv1/beta:0", f32(8)),
                            ("bn_conv1/moving_mean:0", f32(8)), ("bn_conv1/moving_variance:0", np.abs(f32(8)))]))
for i in range(290):
    if i % 3 == 0:
        layers.append((f"activation_{i}", []))
    else:
        layers.append((f"res_dummy{i}", [(f"res_dummy{i}/kernel:0", f32(1, 1, 2, 2)), (f"res_dummy{i}/bias:0", f32(2))]))
layers.append(("rpn_model", [("rpn_conv_shared/kernel:0", f32(3, 3, 8, 16)), ("rpn_conv_shared/bias:0", f32(16)),
                             ("rpn_class_raw/kernel:0", f32(1, 1, 16, 6)), ("rpn_class_raw/bias:0", f32(6)),
                             ("rpn_bbox_pred/kernel:0", f32(1, 1, 16, 12)), ("rpn_bbox_pred/bias:0", f32(12))]))
layers.append(("mrcnn_class_conv1", [("mrcnn_class_conv1/kernel:0", f32(7, 7, 8, 16)), ("mrcnn_class_conv1/bias:0", f32(16))]))
layers.append(("mrcnn_class_logits", [("mrcnn_class_logits/kernel:0", f32(16, 5)), ("mrcnn_class_logits/bias:0", f32(5))]))
layers.append(("mrcnn_mask_deconv", [("mrcnn_mask_deconv/kernel:0", f32(2, 2, 8, 16)), ("mrcnn_mask_deconv/bias:0", f32(8))]))
layers.append(("dtype_zoo", [("dtype_zoo/f64:0", rng.standard_normal((3, 5))), ("dtype_zoo/i32:0", np.arange(-6, 6, dtype=np.int32).reshape(3, 4)),
                             ("dtype_zoo/f16:0", f32(4, 2).astype(np.float16)), ("dtype_zoo/scalar:0", np.float32(2.5)),
                             ("dtype_zoo/u8:0", np.arange(7, dtype=np.uint8))]))

path = os.path.join(HERE, "keras_tiny.h5")
with h5py.File(path, "w") as f:
    f.attrs["layer_names"] = [n.encode("utf8") for n, _ in layers]
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.1.6".encode("utf8")
    for name, weights in layers:
        g = f.create_group(name)
        g.attrs["weight_names"] = [w.encode("utf8") for w, _ in weights]
        for wname, val in weights:
            d = g.create_dataset(wname, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val

expected = {}
for name, weights in layers:
    for wname, val in weights:
        expected[wname.split(":")[0]] = np.asarray(val)
np.savez(os.path.join(HERE, "keras_tiny.npz"), layer_names=np.array([n for n, _ in layers]), **expected)
with open(path, "rb") as src, gzip.GzipFile(path + ".gz", "wb", mtime=0) as dst:
    shutil.copyfileobj(src, dst)
print(os.path.getsize(path), "bytes raw,", os.path.getsize(path + ".gz"), "gzipped")
os.remove(path)
