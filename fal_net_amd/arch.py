"""Layer tables of the three FAL_net variants (reference: models/FAL_netA.py:92-127, FAL_netB.py:92-128,
FAL_netC.py:96-129).  They share one topology -- 7 encoder levels (stride-2 conv + residual block), 6 decoder levels
(nearest-upsample deconv + concat + iconv), a 3x3 logits conv and the 1x1 `conv0` -- and differ only in channel counts,
the attribute name of the backbone in the checkpoint keys, and (A) separable 3x1 / 1x3 residual convs.

Pure Python (no torch): shared by the parameter holders, the launch plan and the seeded-weight generator."""

ARCHS = {
    # enc[i]: channels of encoder level i; dec[lvl]: (deconv Cout, iconv Cout); deconv1 is 64 -> 64 and iconv1 has N outputs
    "B": dict(prefix="backbone", enc=(32, 64, 128, 256, 256, 256, 512),
              dec={6: (256, 256), 5: (128, 256), 4: (128, 256), 3: (128, 128), 2: (64, 64)},
              amask=True, separable=False, no_levels=49, maskr_align_corners=True),
    "C": dict(prefix="synth", enc=(32, 64, 128, 256, 256, 512, 512),
              dec={6: (256, 512), 5: (256, 256), 4: (128, 256), 3: (128, 128), 2: (64, 64)},
              amask=True, separable=False, no_levels=33, maskr_align_corners=True),
    # FAL_netA.py:73-75 separable residual convs; :127 no amask_conv; :264 maskR sampled with grid_sample's default
    # align_corners=False (every other sample in the file passes True)
    "A": dict(prefix="BackBone", enc=(32, 64, 128, 128, 256, 256, 256),
              dec={6: (128, 256), 5: (128, 256), 4: (128, 128), 3: (64, 128), 2: (64, 64)},
              amask=False, separable=True, no_levels=33, maskr_align_corners=False),
}


def conv_shapes(arch, no_levels):
    """Ordered [(state_dict key without .weight/.bias, (Cout, Cin, kh, kw), has_bias)] in the reference's registration order."""
    t = ARCHS[arch]
    p, enc, dec = t["prefix"], t["enc"], t["dec"]
    k1, k2 = ((3, 1), (1, 3)) if t["separable"] else ((3, 3), (3, 3))
    out = []
    for i, ch in enumerate(enc):
        cin = 3 if i == 0 else enc[i - 1] + (1 if i == 1 else 0)
        out.append((f"{p}.conv{i}.0", (ch, cin, 3, 3), True))
        out.append((f"{p}.conv{i}_1.conv1", (ch, ch) + k1, False))
        out.append((f"{p}.conv{i}_1.conv2", (ch, ch) + k2, False))
    below = enc[6]
    for lvl in range(6, 1, -1):
        dch, ich = dec[lvl]
        out.append((f"{p}.deconv{lvl}.conv1", (dch, below, 3, 3), False))
        out.append((f"{p}.iconv{lvl}.0", (ich, dch + enc[lvl - 1], 3, 3), True))
        below = ich
    out.append((f"{p}.deconv1.conv1", (64, below, 3, 3), False))
    out.append((f"{p}.iconv1", (no_levels, 64 + enc[0], 3, 3), False))
    if t["amask"]:
        out.append((f"{p}.amask_conv.0", (48, 96, 3, 3), True))
        out.append((f"{p}.amask_conv.2", (1, 48, 3, 3), False))
    out.append(("conv0", (no_levels, no_levels, 1, 1), True))
    return out
