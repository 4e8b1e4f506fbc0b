#!/bin/bash
out=gpurun_out/r5d; mkdir -p $out
python - > $out/torch_sanity.txt 2>&1 <<'PY'
import torch
x = torch.ones(1 << 20, device="cuda")
print("sum", float(x.sum()))
print("raw", torch._C._cuda_getCurrentRawStream(0), "obj", torch.cuda.current_stream(0).cuda_stream)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print("raw side", torch._C._cuda_getCurrentRawStream(0), "obj", torch.cuda.current_stream(0).cuda_stream, s.cuda_stream)
PY
cat $out/torch_sanity.txt
MI355_RAW_STREAM=0 timeout 200 python -m pytest tests/test_act_storage_gpu.py -m gpu -q -x -k test_cast > $out/cast_objstream.log 2>&1; tail -3 $out/cast_objstream.log
timeout 200 python -m pytest tests/test_act_storage_gpu.py -m gpu -q -x -k test_cast > $out/cast_rawstream.log 2>&1; tail -3 $out/cast_rawstream.log
