#!/usr/bin/env python3
"""CPU simulation (round 6): how many 256 x 128 tiles of a position-ordered level survive the scout's Cauchy-Schwarz test when the
scout stops after `cut` channels -- would a HALF-step scout (32 channels) be enough?  No: on the bench's low-noise clip 510 of 512
tiles stay alive at 32 channels against 32 of 512 (= the tiles whose positions overlap) at 64, the one-step scout that ships.
    python tools/diag/scout_cut_sim.py
corr01 cut 16: 512 / 512   cut 32: 510 / 512   cut 64: 32 / 512   cut 128: 32 / 512
corr002        512           270                 32                  32
corr05         512           512                 512                 510
smooth         512           173                 70                  64"""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidtome_amd import sites
torch.manual_seed(0)
g=torch.Generator().manual_seed(0)
B,F,N,C=1,4,4096,320
for regime in ("corr01","corr002","corr05","smooth"):
    x=sites.regime_tokens(regime,B,F,N,C,g)            # (B,F,N,C)
    x=torch.nn.functional.layer_norm(x,(C,)).half().float()
    xh=x/x.norm(dim=-1,keepdim=True)
    hi=(1024*xh).half().float()/1024                   # filter operand (fp16 of scaled xhat)
    src=hi[0,0]; dst=hi[0,1]                            # frame 0 = src rows, frame 1 = dst frame (position-ordered both)
    full=src@dst.T
    seed=torch.diagonal(full).clone()-1e-4              # same-position guess
    W=2.4e-3
    for cut in (16,32,64,128):
        part=src[:,:cut]@dst[:,:cut].T
        ra=src[:,cut:].norm(dim=1)*(1+2**-16); rb=dst[:,cut:].norm(dim=1)*(1+2**-16)
        rbt=rb.view(-1,128).max(dim=1).values           # per dst tile
        alive=0; tot=0
        for st in range(0,N,256):
            need=(seed[st:st+256]-W)[:,None]
            for jt in range(N//128):
                bound=part[st:st+256, jt*128:(jt+1)*128]+ra[st:st+256,None]*rbt[jt]
                tot+=1; alive+=bool((bound>=need).any())
        print(regime,'cut',cut,'tiles alive',alive,'/',tot, 'ideal (overlapping positions)', (N//256)*2)
