import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'geo-trax_amd')); sys.path.insert(0,ROOT)
from geotrax_amd import _lib, ops
ctx=_lib.default_context(0)
# usage: one_layer.py NB H cin cout k s iters
NB,H,cin,cout,k,s,it=[int(x) for x in sys.argv[1:8]]
ms,fl=ops.conv2d_time(int(os.environ.get("GTX_DT","0")),NB,H,H,cin,cout,k,s,iters=it,ctx=ctx)
print(f"{ms*1000:.1f} us {fl/ms/1e9:.1f} TF/s")
