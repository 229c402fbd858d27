import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""Print the channelizer planner's pass structure for a bench-shaped bank:  SDRX_CHAN_DEBUG=1 python tools/plan_debug.py 256   (needs a GPU: the bank allocates its buffers)"""
import sdrangel_amd as sa, numpy as np, sys
n_ch=int(sys.argv[1])
k=np.arange(n_ch)
fcs=(-25_000_000 + k*(50_000_000/255)).astype(np.int64).tolist() if n_ch==256 else (-15_000_000 + k*(30_000_000/(n_ch-1)) + 137*k).astype(np.int64).tolist()
h=sa.ChannelizerBank(61_440_000,[48000]*n_ch,fcs)
