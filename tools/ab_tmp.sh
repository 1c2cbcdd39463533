for v in tall1 tall2; do for dt in f32 f16; do echo "== $v $dt"; MRCNN_HIP_LIB=$PWD/mask-rcnn-coreml_amd/libvar_$v.so python tools/conv_microbench.py 10 $dt 2>&1 | grep -v amdgpu; done; done
MRCNN_HIP_LIB=$PWD/mask-rcnn-coreml_amd/libvar_tall2.so timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "resnet101_256 or fp16_small" 2>&1 | tail -3
