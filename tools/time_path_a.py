#!/usr/bin/env python3
"""Stage timing of RobustMVD.forward at a BASELINE config, NCHW vs channels_last for the MIOpen part."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
model, _ = BN.build_robustmvd(dev)
s = BN.adapted_sample(model, 0, H, W, V)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def run(cl):
    imgs = [im.contiguous(memory_format=torch.channels_last) if cl else im for im in s["images"]]
    m = model.to(memory_format=torch.channels_last) if cl else model.to(memory_format=torch.contiguous_format)
    with torch.no_grad():
        for rep in range(4):
            e0 = ev()
            allk, enc_key = m.encoder(imgs[0])
            feats = m.encoder.conv3(m.encoder.conv2(m.encoder.conv1(torch.cat(imgs[1:], 0))))
            ctx = m.context_encoder(enc_key)
            e1 = ev()
            corrs, masks, _ = m.corr_block(feat_key=enc_key, intrinsics_key=s["intrinsics"][0], feat_sources=list(torch.split(feats, 1, 0)),
                                           source_to_key_transforms=s["poses"][1:], intrinsics_sources=s["intrinsics"][1:],
                                           num_sampling_points=256, min_depth=0.4, max_depth=1000.0)
            e2 = ev()
            fused, _ = m.fusion_block(corrs=corrs, masks=masks)
            e3 = ev()
            if cl: fused = fused.contiguous(memory_format=torch.channels_last)
            allf, encf = m.fusion_enc_block(corr=fused, ctx=ctx)
            e4 = ev()
            dec = m.decoder(enc_fused=encf, all_enc={**allk, **allf})
            e5 = ev()
            torch.cuda.synchronize()
    print(("channels_last" if cl else "nchw         "), "encoders %.2f  K1 %.2f  fusion %.2f  cv-encoder %.2f  decoder %.2f  total %.2f ms" % (
        e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4), e4.elapsed_time(e5), e0.elapsed_time(e5)), flush=True)
run(False); run(True)
