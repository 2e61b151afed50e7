"""CPU check of the HOST side of the split-operand arithmetic (vps_amd/nhwc.py:PackedConv packs on whatever device it is given):
the operand planes, the per-channel power-of-two scaling of f16x3, the k ordering (tap-major / chunk-major), the parity classes of
the transposed convs and the MFMA-fragment permutation are undone from the packed buffer exactly as the kernels index them
(include/vps_hip.h: w_split [plane][class][cout_pad/32][kpad/16][lane = 32*((k/8)%2) + cout%32][8]), the activation split of
conv_mfma.hip:split_act is mirrored in torch, the product table of Split<MODE> is applied, and the result — accumulated in float64,
i.e. the arithmetic DESIGN, not the accumulation order of the matrix pipe — is compared with the exact float64 convolution:

    f16x3  : |err| <= 3 * 2^-22 * sum|x||w|     (the GPU test measures the same bound through the kernels)
    bf16x6 : |err| <= 2^-22 * sum|x||w|
    bf16x3 : |err| <= 2^-15 * sum|x||w|
    bf16   : |err| <= 2^-7  * sum|x||w|
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vps_amd import hip, nhwc

PA = {hip.PREC_F16X3: [0, 1, 0], hip.PREC_BF16X6: [0, 0, 1, 1, 0, 2], hip.PREC_BF16X3: [0, 0, 1], hip.PREC_BF16: [0]}
PB = {hip.PREC_F16X3: [1, 2, 0], hip.PREC_BF16X6: [0, 1, 0, 1, 2, 0], hip.PREC_BF16X3: [0, 1, 0], hip.PREC_BF16: [0]}
BOUND = {hip.PREC_F16X3: 3 * 2.0 ** -22, hip.PREC_BF16X6: 2.0 ** -22, hip.PREC_BF16X3: 2.0 ** -15, hip.PREC_BF16: 2.0 ** -7}


def unpack_planes(pc):
    """w_split -> float64 [plane][class][cout_pad][tap][cin_pad] in natural order (undoes fragment order and k order)"""
    ws = pc.w_split
    if not (pc.deform and pc.korder == 0):
        P, C, OB, KB, two, thirty2, eight = ws.shape
        assert (two, thirty2, eight) == (2, 32, 8)
        ws = ws.permute(0, 1, 2, 5, 3, 4, 6).reshape(P, C, OB * 32, KB * 16)
    ws = ws.double()
    P, C, O, K = ws.shape
    ntap = pc.KH * pc.KW
    if pc.korder == 0:
        return ws[..., :ntap * pc.cin_pad].reshape(P, C, O, ntap, pc.cin_pad)
    nch = (pc.cin_pad + 31) // 32
    return ws[..., :nch * ntap * 32].reshape(P, C, O, nch, ntap, 32).permute(0, 1, 2, 4, 3, 5).reshape(P, C, O, ntap, nch * 32)


def split_act(x, prec):
    """mirror of conv_mfma.hip:split_act -> list of float64 planes whose weighted sum is the staged activation"""
    if prec == hip.PREC_F16X3:
        h0 = x.to(torch.float16)
        h1 = ((x - h0.float()) * 2048.0).to(torch.float16)
        return [h0.double(), h1.double()]
    n = {hip.PREC_BF16X6: 3, hip.PREC_BF16X3: 2, hip.PREC_BF16: 1}[prec]
    r, out = x.clone(), []
    for _ in range(n):
        h = r.to(torch.bfloat16)
        out.append(h.double())
        r = r - h.float()
    return out


def packed_planes_checks(pc, prec):
    planes = unpack_planes(pc)
    if prec == hip.PREC_F16X3:
        g0, g1, g2 = planes
        # g2 = 2^-11 g0: exact while it stays a normal fp16 number, i.e. for |g0| >= 2^-3 (weights less than 2^14 below their
        # channel's maximum); below, the fp16 rounding of the product (what the host stores) - an error <= 2^-25 per element
        assert torch.equal(g2, (g0 * 2.0 ** -11).to(torch.float16).double())
        big = g0.abs() >= 0.125
        assert torch.equal(g2[big], (g0 * 2.0 ** -11)[big])
        top = g0.abs().amax(dim=(0, 2, 3))[:pc.cout]
        assert bool(((top >= 2.0 ** 11) & (top <= 2.0 ** 12)).all())                  # every output channel fills the fp16 significand range
        sc = pc.scale if pc.scale is not None else torch.ones(pc.cout)
        assert bool((torch.frexp(sc.float())[0].abs() == 0.5).all()) or pc.has_scale  # pure powers of two unless a BN scale is folded in
    return planes


@pytest.mark.parametrize('prec', [hip.PREC_F16X3, hip.PREC_BF16X6, hip.PREC_BF16X3, hip.PREC_BF16], ids=['f16x3', 'bf16x6', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('cin,cout,k,stride,pad', [(64, 96, 3, 1, 1), (12, 40, 3, 1, 1), (82, 130, 5, 2, 2), (256, 64, 1, 1, 0)],
                         ids=['chunk_major', 'tap_major_small_cin', 'ragged_chunk_5x5_s2', 'pointwise'])
def test_packed_conv_operands_reproduce_the_convolution(prec, cin, cout, k, stride, pad):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(1, cin, 9, 11, generator=g) * torch.logspace(-2, 2, cin).view(1, cin, 1, 1)      # 4 decades of activation range
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5 * torch.logspace(-3, 1, cout).view(cout, 1, 1, 1)
    pc = nhwc.PackedConv(w, None, None, stride, pad, device=torch.device('cpu'), prec=prec)
    assert pc.prec == prec and pc.nclass == 1
    planes = packed_planes_checks(pc, prec)
    xa = split_act(x, prec)
    if prec == hip.PREC_F16X3:
        xa = [xa[0], xa[1]]                                                       # x = h0 + 2^-11 h1; g2 carries the 2^-11
    y = torch.zeros(1, cout, *F.conv2d(x, w, stride=stride, padding=pad).shape[2:], dtype=torch.float64)
    for pa, pb in zip(PA[prec], PB[prec]):
        wp = planes[pb][0, :cout, :, :cin].reshape(cout, k, k, cin).permute(0, 3, 1, 2)      # [cout][cin][ky][kx]
        y += F.conv2d(xa[pa], wp, stride=stride, padding=pad)
    if pc.scale is not None:
        y = y * pc.scale.double().view(1, -1, 1, 1)
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    den = F.conv2d(x.double().abs(), w.double().abs(), stride=stride, padding=pad)
    rel = float(((y - ref).abs() / den).max())
    print('prec %d %s: max |err| / sum|x||w| = %.3e (bound %.3e)' % (prec, (cin, cout, k, stride), rel, BOUND[prec]))
    assert rel <= BOUND[prec]
    # pad rows / columns of the packed buffer are zero (the kernels contract over cin_pad and kpad)
    assert float(planes[0][0, cout:].abs().max() if planes[0].shape[1] > cout else 0.0) == 0.0
    assert float(planes[0][0, :, :, cin:].abs().max() if planes[0].shape[3] > cin else 0.0) == 0.0


@pytest.mark.parametrize('prec', [hip.PREC_F16X3, hip.PREC_BF16X6], ids=['f16x3', 'bf16x6'])
def test_packed_transposed_conv_classes_reproduce_conv_transpose(prec):
    """ConvTranspose2d 4x4 s2 p1 (the FlowNet2 deconvs) = 4 output-parity classes of 2x2 stride-1 convs: the class weights and
    per-class paddings PackedConv derives, applied with the split operands, give F.conv_transpose2d"""
    cin, cout = 64, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, cin, 7, 9, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) * 0.05
    pc = nhwc.PackedConv(w, None, None, 2, 1, transposed=True, device=torch.device('cpu'), prec=prec)
    assert pc.nclass == 4 and (pc.KH, pc.KW) == (2, 2)
    planes = unpack_planes(pc)
    xa = split_act(x, prec)
    ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
    den = F.conv_transpose2d(x.double().abs(), w.double().abs(), stride=2, padding=1)
    y = torch.zeros_like(ref)
    H, W = x.shape[2:]
    for py in range(2):
        for px in range(2):
            cls = py * 2 + px
            acc = torch.zeros(1, cout, H, W, dtype=torch.float64)
            for pa, pb in zip(PA[prec], PB[prec]):
                wp = planes[pb][cls, :cout, :, :cin].reshape(cout, 2, 2, cin).permute(0, 3, 1, 2)
                # tap (uy, ux) of class (py, px) reads input (q + u - pad[class]) : pad the input so that a valid conv does it
                ty, tx = pc.pad_y[py], pc.pad_x[px]
                xp = F.pad(xa[pa], (tx, 1 - tx, ty, 1 - ty))
                acc += F.conv2d(xp, wp)
            if pc.scale is not None:
                acc = acc * pc.scale.double().view(1, -1, 1, 1)
            y[:, :, py::2, px::2] = acc
    rel = float(((y - ref).abs() / den).max())
    assert rel <= BOUND[prec], rel


def test_n16_weight_fragment_gather_from_the_common_packed_layout():
    """conv_mfma_n16_kernel (16x16x32 MFMA, 5..16 output channels) takes its A operand — 16 output channels x 32 k — from the SAME
    packed buffer as every other kernel: lane l of the fragment of 32-k step s reads the 16 bytes of lane (l & 15) + 32 (g & 1) of
    16-k step 2s + (g >> 1), g = l >> 4 (csrc/conv_mfma.hip: load_W / `wlane`). Mirror that address arithmetic on the host and check
    that the lane then holds W[channel l & 15][k = 32 s + 8 g .. + 7], the operand layout of v_mfma_f32_16x16x32_f16, for every plane
    and k step of a chunk-major 3x3 layer and of a transposed layer's parity classes."""
    g = torch.Generator().manual_seed(3)
    for transposed in (False, True):
        w = torch.randn((82, 16, 4, 4) if transposed else (16, 82, 3, 3), generator=g)
        pc = nhwc.PackedConv(w, None, None, stride=2 if transposed else 1, padding=1, transposed=transposed, device='cpu', prec=hip.PREC_F16X3)
        assert pc.korder == 1 and pc.cout_pad == 32
        ws = pc.w_split                                        # [plane][class][cout_pad/32][kpad/16][2][32][8]
        P, C, OB, KB = ws.shape[:4]
        flat = ws.reshape(P, C, OB, KB, 512)                   # one 1 KB fragment of the 32x32x16 shape = 512 elements, lane-major
        nat = ws.permute(0, 1, 2, 5, 3, 4, 6).reshape(P, C, OB * 32, KB * 16)     # natural [cout][k]
        lanes = torch.arange(64)
        grp = lanes >> 4
        src_lane = (lanes & 15) + 32 * (grp & 1)
        for s in range(KB // 2):
            for p in range(P):
                for c in range(C):
                    frag = torch.stack([flat[p, c, 0, 2 * s + int(grp[l] >> 1), int(src_lane[l]) * 8:int(src_lane[l]) * 8 + 8] for l in range(64)])   # [64 lanes][8]
                    want = torch.stack([nat[p, c, int(l & 15), 32 * s + 8 * int(grp[l]):32 * s + 8 * int(grp[l]) + 8] for l in lanes])
                    assert torch.equal(frag, want), (transposed, s, p, c)
        assert float(nat[:, :, 16:].abs().max()) == 0.0       # channels 16..31 of the padded block are zero: one 16-row block suffices


@pytest.mark.parametrize('cin,k,stride,PH', [(6, 3, 1, 8), (11, 3, 1, 8), (3, 7, 2, 8), (12, 7, 2, 4)], ids=['6to64_3x3', '11to64_3x3', '3to64_7x7s2', '12to64_7x7s2'])
def test_thin_kernel_indexing_reproduces_the_convolution(cin, k, stride, PH):
    """csrc/conv_thin.hip on the CPU: the block's input patch staged pixel-major with cin_pad channels per pixel, the k order (kernel
    row j, position within the row's KS*C4 consecutive values, padded to 16), the lane's 8-k fragment = 8 consecutive staged values
    at (row*S + j, column*S), the weight fragment gathered from `PackedConv.w_thin` exactly as the kernel indexes it, the padding k of
    a row reading the NEXT pixels' values against zero weights - with both operands split as in f16x3 - against the float64 conv."""
    S, KS, cout = stride, k, 64
    g = torch.Generator().manual_seed(cin + k)
    H, W = 21, 45                                             # patches overhang right and bottom; 2 x 2 (x 3 for PH = 4) blocks
    x = torch.randn(1, cin, H, W, generator=g) * torch.logspace(-1, 1, cin).view(1, cin, 1, 1)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5 * torch.logspace(-2, 1, cout).view(cout, 1, 1, 1)
    pc = nhwc.PackedConv(w, None, None, stride, k // 2, device=torch.device('cpu'), prec=hip.PREC_F16X3)
    assert pc.w_thin is not None and pc.korder == 0
    C4 = pc.cin_pad
    RL = KS * C4
    NK16 = (RL + 15) // 16
    assert tuple(pc.w_thin.shape) == (KS, 2, NK16, 2, 2, 32, 8)
    wt = pc.w_thin.double().reshape(KS, 2, NK16, 2, 64, 8)                      # [j][plane][s][nb][lane][e]
    g0, g1 = wt[:, 0], wt[:, 1]
    g2 = (g0 * 2.0 ** -11).to(torch.float16).double()                           # derive_weight_plane
    wpl = [g0, g1, g2]
    Ho, Wo = pc.out_hw(H, W)
    pad = k // 2
    PR, PC = (PH - 1) * S + KS, 31 * S + KS
    TM = PH // 4
    xa = torch.zeros(1, C4, H, W); xa[:, :cin] = x
    xs = split_act(xa, hip.PREC_F16X3)                                          # planes h0, h1 [1][C4][H][W]
    y = torch.zeros(cout, Ho, Wo, dtype=torch.float64)
    lane = torch.arange(64)
    p32, kh = lane & 31, lane >> 5
    for ty in range((Ho + PH - 1) // PH):
        for tx in range((Wo + 31) // 32):
            iy0, ix0 = ty * PH * S - pad, tx * 32 * S - pad
            As = []
            for pl in range(2):
                a = torch.zeros(PR * PC * C4 + 16, dtype=torch.float64)
                for r in range(PR):
                    for c in range(PC):
                        iy, ix = iy0 + r, ix0 + c
                        if 0 <= iy < H and 0 <= ix < W:
                            a[(r * PC + c) * C4:(r * PC + c + 1) * C4] = xs[pl][0, :, iy, ix]
                As.append(a)
            for wave in range(4):
                for a_ in range(TM):
                    acc = torch.zeros(2, 32, 32, dtype=torch.float64)          # [nb][cout % 32][pixel]
                    for j in range(KS):
                        for s in range(NK16):
                            base = (((wave * TM + a_) * S + j) * PC + p32 * S) * C4 + s * 16 + kh * 8        # per lane
                            idx = base[:, None] + torch.arange(8)[None]
                            af = [As[0][idx], As[1][idx]]                       # [lane][8]
                            for pa, pb in zip(PA[hip.PREC_F16X3], PB[hip.PREC_F16X3]):
                                for nb in range(2):
                                    wf = wpl[pb][j, s, nb]                      # [lane][8]: lane = 32 * kh + cout % 32
                                    # D[co][px] += sum over the two k halves and the 8 values of a lane
                                    for h in range(2):
                                        acc[nb] += wf[32 * h:32 * h + 32] @ af[pa][32 * h:32 * h + 32].t()
                    oy = ty * PH + wave * TM + a_
                    if oy >= Ho:
                        continue
                    for px in range(32):
                        ox = tx * 32 + px
                        if ox < Wo:
                            y[:, oy, ox] = acc[:, :, px].reshape(64)
    y = y * pc.scale.double().view(-1, 1, 1)
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad)[0]
    den = F.conv2d(x.double().abs(), w.double().abs(), stride=stride, padding=pad)[0]
    rel = float(((y - ref).abs() / den).max())
    print('thin %s: max |err| / sum|x||w| = %.3e' % ((cin, k, stride, PH), rel))
    assert rel <= BOUND[hip.PREC_F16X3]
