"""Export of a ResNet (net.py here, or the reference's net.py -- same parameter names) into the
flat weight blob the engine's MFMA kernels consume (format: csrc/fpc_nn.h "weight blob").

eval()-mode BatchNorm is folded into the preceding conv (the search runs under model.eval() +
torch.no_grad(), alphazero.py:262 / mcts.py:15):   w' = w * g/sqrt(var+eps),  b' = (b-mean)*g/sqrt(var+eps)+beta.
Conv weights go to [tap][Cout_pad][Cin_pad] (K contiguous); the policy Linear's input axis is
permuted from the reference's NCHW flatten (ch*R*R + pos) to the engine's NHWC flatten
(pos*A_ch + ch) and both axes are zero-padded to tile multiples."""
import os
import struct

import numpy as np
import torch


def _cpu(t):
    """a detached fp32 host copy; the caller's module stays on its device"""
    return t.detach().to("cpu", torch.float32)


def _fold(conv, bn):
    w = _cpu(conv.weight)
    b = _cpu(conv.bias) if conv.bias is not None else torch.zeros(w.shape[0])
    s = _cpu(bn.weight) / torch.sqrt(_cpu(bn.running_var) + bn.eps)
    return w * s.view(-1, 1, 1, 1), (b - _cpu(bn.running_mean)) * s + _cpu(bn.bias)


def _to16(t, dtype):
    t = t.contiguous()
    if dtype == 0:
        return t.to(torch.bfloat16).view(torch.int16).numpy().tobytes()
    return t.to(torch.float16).view(torch.int16).numpy().tobytes()


def _conv_section(w, b, cin_pad, cout_pad, dtype):
    cout, cin = w.shape[0], w.shape[1]
    wt = torch.zeros(9, cout_pad, cin_pad)
    wt[:, :cout, :cin] = w.permute(2, 3, 0, 1).reshape(9, cout, cin)     # tap = ky*3+kx
    bb = torch.zeros(cout_pad)
    bb[:cout] = b
    return [_to16(wt, dtype), bb.numpy().astype(np.float32).tobytes()]


# Fragment order / block tile of the policy Linear: 2 = 16x16x32 fragments over 384-column groups (k_fcw: 256 x 384 block
# tiles, ONE round of blocks; the default wherever its one-round K-split exists on a 256-CU part: every board of 8..14
# a side), 1 = the same fragment order over 256-column groups (k_fc16: long and short blocks in two rounds; the fallback).
# FPC_FC_LAYOUT forces one.  (0, the 32x32x16 order of k_fc, was retired in round 5 together with blob version 2.)
FC_LAYOUT = int(os.environ["FPC_FC_LAYOUT"]) if "FPC_FC_LAYOUT" in os.environ else None


def fcw_split(R, cus=256):
    """K-splits k_fcw would use for an R x R board on a part with `cus` CUs (csrc/fpc_nn.h: plan_fcw), 0 if none"""
    A = (8 * R + 8) * R * R
    groups, stages = (A + 383) // 384, ((A + 511) // 512 * 512) // 64
    best = 0
    for sk in range(1, 9):
        if groups * sk <= cus and stages % sk == 0 and stages // sk >= 4:
            best = sk
    return best


def default_fc_layout(R):
    if FC_LAYOUT is not None:
        return FC_LAYOUT
    return 2 if fcw_split(R) else 1


def export_weights(model, dtype=0, fc_layout=None):
    """model: ResNet in eval semantics.  dtype 0 = bf16, 1 = fp16.  Returns bytes."""
    F = model.startBlock[0].weight.shape[0]
    nblocks = len(model.backBone)
    A_ch = model.policyHead[0].weight.shape[0]
    fc = model.policyHead[4]
    A = fc.weight.shape[0]
    RR = A // A_ch
    R = int(round(RR ** 0.5))
    assert R * R == RR and fc.weight.shape[1] == A
    fc_layout = default_fc_layout(R) if fc_layout is None else int(fc_layout)
    gw = 384 if fc_layout == 2 else 256
    Np = (A + gw - 1) // gw * gw         # a block owns gw columns
    Kp = (A + 511) // 512 * 512      # k_fc: K/16 k-steps, split-K 4 (8 for the short blocks), 4 k-steps per stage, stages in pairs
    secs = []
    w, b = _fold(model.startBlock[0], model.startBlock[1])
    Fp = (F + 127) // 128 * 128
    secs += _conv_section(w, b, 32, Fp, dtype)
    for blk in model.backBone:
        w, b = _fold(blk.conv1, blk.bn1)
        secs += _conv_section(w, b, F, Fp, dtype)
        w, b = _fold(blk.conv2, blk.bn2)
        secs += _conv_section(w, b, F, Fp, dtype)
    w, b = _fold(model.policyHead[0], model.policyHead[1])
    secs += _conv_section(w, b, F, 128, dtype)
    w, b = _fold(model.valueHead[0], model.valueHead[1])
    secs += _conv_section(w, b, F, 128, dtype)
    # policy Linear: in index ch*RR+pos -> pos*A_ch+ch ; pad to [Np][Kp]
    fw = _cpu(fc.weight).view(A, A_ch, RR).permute(0, 2, 1).reshape(A, A)
    t16 = torch.bfloat16 if dtype == 0 else torch.float16
    fwp = torch.zeros(Np, Kp, dtype=t16)
    fwp[:A, :A] = fw.to(t16)
    del fw
    if fc_layout not in (1, 2):
        raise ValueError("fc_layout must be 1 (k_fc16) or 2 (k_fcw); 0 (k_fc) was retired")
    # MFMA fragment order of k_fc16 / k_fcw (16x16x32): [kstep32][n_tile16][lane = 16*q + c][8],
    # element (ks, nt, q, c, e) = W'[nt*16 + c][ks*32 + q*8 + e]
    wf = fwp.view(Np // 16, 16, Kp // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()
    del fwp
    secs.append(wf.view(torch.int16).numpy().tobytes())
    del wf
    fb = np.zeros(Np, np.float32)
    fb[:A] = _cpu(fc.bias).numpy()
    secs.append(fb.tobytes())
    vfc = model.valueHead[4]
    vw = torch.zeros(RR, 32)
    vw[:, :24] = _cpu(vfc.weight).view(24, RR).t()
    secs.append(vw.numpy().astype(np.float32).tobytes())
    secs.append(struct.pack("<f", float(_cpu(vfc.bias).item())))
    # version 3: the header word fc_layout names the policy Linear's layout (version 2, rounds 1-2: k_fc's 32x32x16
    # order without that word; an engine of that time refuses version 3, this engine refuses version 2)
    version = 3
    out = bytearray(struct.pack("<4s9i24x", b"FPCW", version, R, F, nblocks, dtype, A_ch, Np, Kp, fc_layout))
    assert len(out) == 64
    for s in secs:
        out += b"\0" * ((-len(out)) % 64)
        out += s
    return bytes(out)
