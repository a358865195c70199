#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04h
{
REPS=1 bash tools/exp/run_matrix.sh \
 "x4 seq|x4|MI355_PIPELINE=0|" \
 "x4 no general|x4|MI355_PIPELINE=0 MI355_XDEBUG=1|" \
 "x4 no fast|x4|MI355_PIPELINE=0 MI355_XDEBUG=2|" 
python - <<'PY'
import numpy as np, torch, ctypes as C
from cudavideostream_amd import CUDACore, synth, lib
w,h,T=1920,1080,8
base, frames = synth.webcam_stream(T, w, h, seed=21)
dev=torch.device("cuda",0)
with CUDACore(w,h,max_batch=T,sample_mat_data=base) as core:
    core.use_torch_stream()
    d=torch.from_numpy(frames).to(dev); n=3*w*h; cap=T*n//4
    off=torch.zeros(T+1,dtype=torch.int32,device=dev); xs=torch.empty(cap,dtype=torch.int32,device=dev); df=torch.empty(cap,dtype=torch.uint8,device=dev)
    core.diff_stream_batch(d,T,off,xs,df,cap); torch.cuda.synchronize()
    print("offsets", off.cpu().numpy())
PY
} > gpurun_out/r04h/log.txt 2>&1
cat gpurun_out/r04h/log.txt
