import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames
mode = sys.argv[1]
spec, punct = "ar4ja:1/2:1024", "1,1,1,1,0"
msgs, llrs, full = awgn_frames(spec, 6, 2.3, 21, punct)
dec = lt.LdpcDecoder(alist(spec), "Minsumf32", punct)
if mode == "lat0":
    dec.set("latency", 0)
if mode != "none":
    for b in range(6):
        dec.decode(llrs[b], 40)
        if mode != "f32only":
            dec.decode(llrs[b].astype(np.float64), 40)
if mode == "del":
    del dec
import torch
print(mode, "torch sees", torch.cuda.device_count(), "devices; init:", end=" ")
x = torch.zeros(4).cuda()
print("ok", x.device)
