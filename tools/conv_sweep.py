"""Per-layer timing of the YOLOv8s convolution shapes at a given input size and batch, each layer alone on the GPU
(gtx_op_conv2d_time: HIP events around 20 back-to-back launches).
Usage: python tools/conv_sweep.py [imgsz] [batch] [all|k3s1|k1|k3s2] [f16|f32|f32s ...]   (f32s = split-f16x3)
Output committed as profiles/rNN_conv_layer_sweep_<dtype>_b<batch>.txt."""
import sys
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'geo-trax_amd')); sys.path.insert(0,ROOT)
import numpy as np
from geotrax_amd import _lib, ops
ctx=_lib.default_context(0)
S=int(sys.argv[1]) if len(sys.argv)>1 else 1920
NB=int(sys.argv[2]) if len(sys.argv)>2 else 1
ONLY=sys.argv[3] if len(sys.argv)>3 and sys.argv[3]!='all' else ''
DTS=[a for a in sys.argv[4:]] or ['f16']
DTID={'f16':0,'f32':1,'f32s':2}
HW={2:S//2,4:S//4,8:S//8,16:S//16,32:S//32}
# (name, cin, cout, k, stride, input-level-stride)
L=[("m1",32,64,3,2,2),("m2.cv1",64,64,1,1,4),("m2.m",32,32,3,1,4),("m2.cv2",96,64,1,1,4),("m3",64,128,3,2,4),
   ("m4.cv1",128,128,1,1,8),("m4.m",64,64,3,1,8),("m4.cv2",256,128,1,1,8),("m5",128,256,3,2,8),
   ("m6.cv1",256,256,1,1,16),("m6.m",128,128,3,1,16),("m6.cv2",512,256,1,1,16),("m7",256,512,3,2,16),
   ("m8.cv1",512,512,1,1,32),("m8.m",256,256,3,1,32),("m8.cv2",768,512,1,1,32),("m9.cv1",512,256,1,1,32),("m9.cv2",1024,512,1,1,32),
   ("m12.cv1",768,256,1,1,16),("m12.cv2",384,256,1,1,16),("m15.cv1",384,128,1,1,8),("m15.cv2",192,128,1,1,8),
   ("m16",128,128,3,2,8),("m18.cv1",384,256,1,1,16),("m19",256,256,3,2,16),("m21.cv1",768,512,1,1,32),
   ("h0.s1",128,192,3,1,8),("h1.s1",256,192,3,1,16),("h2.s1",512,192,3,1,32),("h0.s2c",128,128,3,1,8),("h0.s2b",64,64,3,1,8)]
for dt,name in [(DTID[d],d) for d in DTS]:
    tot=0;totf=0
    for (nm,cin,cout,k,s,lv) in L:
        h=HW[lv]
        if ONLY and not ((ONLY=='k3s1' and k==3 and s==1) or (ONLY=='k1' and k==1) or (ONLY=='k3s2' and s==2)): continue
        ms,fl=ops.conv2d_time(dt,NB,h,h,cin,cout,k,s,iters=20,ctx=ctx)
        tot+=ms
        es=2 if dt==0 else 4
        by=NB*(h*h*cin+(h//s)*(h//s)*cout)*es
        print(f"{name} {nm:8s} {cin:4d}->{cout:4d} k{k} s{s} {h:4d}^2  {ms*1000:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e6:7.0f} GB/s")

    print(f'TOTAL {tot*1000:.1f} us')
