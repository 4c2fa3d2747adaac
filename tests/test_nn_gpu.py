"""The MFMA ResNet (csrc/fpc_nn.h) against a plain PyTorch fp32 CPU reference of the same
architecture (net.py mirrors /root/reference/src/py/net.py:6-63), and the fused device-resident
search loop against the parity-checked step-wise path."""
import ctypes as C
import random

import numpy as np
import pytest

import fpc_ffi
from fpc_testlib import gold, make_engine, run_external_search

pytestmark = pytest.mark.gpu


class Spec:
    """the gameType attributes net.ResNet reads (wrapper.cpp:176-181, :209-210)"""
    def __init__(self, R):
        self.R = R
        self.num_state_channels = 24
        self.num_action_channels = 8 * R + 8
        self.action_space_size = self.num_action_channels * R * R
        self.state_space_size = 24 * R * R

    def nRows(self):
        return self.R

    def nCols(self):
        return self.R


def _model(R, blocks, hidden, seed=0):
    import torch
    import net
    torch.manual_seed(seed)
    m = net.ResNet(Spec(R), blocks, hidden, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():                      # non-trivial BN statistics so that the fold is exercised
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    return m.eval()


INV_OF = {8: 2, 9: 2, 10: 2, 11: 3, 12: 3, 13: 3, 14: 3}


@pytest.fixture(autouse=True)
def _dev_knobs(monkeypatch):
    """the engine consults its developer knobs (FPC_TOWERW, FPC_TOWER_WAVES, FPC_TOWERW_ROWS, FPC_NO_TOWER) only when
    FPC_DEV_KNOBS=1: the bit-identity tests of this file opt in, a product run never does"""
    monkeypatch.setenv("FPC_DEV_KNOBS", "1")


def _positions(R, n):
    if R not in (8, 14):          # no reference build at this size: random legal playouts on the engine itself
        import random
        import positions
        from engine_cases import synthetic_entries
        turn, entries = positions.start_entries(R) if R in (10, 13) else synthetic_entries(R, INV_OF[R])
        eng = make_engine("gpu", R, INV_OF[R], max_games=4, max_sims=4)
        rng = random.Random(R)
        b = fpc_ffi.board_from_dict(R, turn, entries)
        out = []
        while len(out) < n:
            lm = eng.legal_moves([b])[0]
            if not lm or len(out) % 9 == 8:
                b = fpc_ffi.board_from_dict(R, turn, entries)
                lm = eng.legal_moves([b])[0]
            b = eng.take_action([b], [lm[rng.randrange(len(lm))][2]])[0]
            out.append(fpc_ffi.clone_board(b))
        eng.close()
        return out
    g = gold(R)
    out = []
    for stride in (7, 3, 1):                     # a denser sweep of the recorded playouts when more positions are asked for
        out = []
        for game in g["playouts"]:
            for rec in game[::stride]:
                out.append(fpc_ffi.board_from_lists(R, rec["before"]["turn"], rec["before"]["pl"]))
                if len(out) == n:
                    return out
    return out


@pytest.mark.parametrize("R,blocks,hidden,dtype,tol", [(8, 4, 64, 1, 1e-3), (8, 4, 64, 0, 8e-3), (8, 2, 128, 1, 1e-3),
                                                       (14, 2, 64, 1, 1e-3), (14, 2, 64, 0, 8e-3),
                                                       (14, 3, 128, 1, 1e-3), (14, 3, 128, 0, 8e-3), (8, 3, 128, 0, 8e-3),
                                                       (8, 2, 256, 1, 1e-3), (10, 2, 128, 1, 1e-3),
                                                       # hidden 256: k_towerw on one wave row at every row-tile count (MT = 4, 6, 7, 8, 9, 11, 13)
                                                       (8, 3, 256, 0, 8e-3), (9, 2, 256, 1, 1e-3), (10, 2, 256, 1, 1e-3), (11, 2, 256, 1, 1e-3), (12, 2, 256, 1, 1e-3),
                                                       (13, 2, 256, 1, 1e-3), (14, 2, 256, 1, 1e-3), (14, 2, 256, 0, 8e-3),
                                                       # sizes without a reference layout (hidden 128 off 14x14: k_towerw, MT 3 / 4 / 5 / 6)
                                                       (9, 2, 128, 1, 1e-3), (11, 2, 128, 1, 1e-3), (12, 2, 128, 1, 1e-3), (13, 2, 128, 1, 1e-3)])
def test_resnet_forward_vs_torch_fp32(R, blocks, hidden, dtype, tol):
    """north_star tolerance: policy/value logits within 1e-3 of the fp32 reference.  Met with fp16
    MFMA operands; bf16 (8 mantissa bits) is reported with its own, looser, bound."""
    import torch
    import weights
    m = _model(R, blocks, hidden)
    eng = make_engine("gpu", R, INV_OF[R], max_games=40, max_sims=4, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(m, dtype))
    boards = _positions(R, 37)
    n = len(boards)
    enc = np.concatenate([eng.encode([b]) for b in boards])          # per-position rotation
    with torch.no_grad():
        ref_l, ref_v = m(torch.from_numpy(enc))
    x = torch.from_numpy(enc).cuda()
    lg = torch.empty(n, eng.A, device="cuda")
    va = torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    eng.nn_forward(x.data_ptr(), n, lg.data_ptr(), va.data_ptr())
    el = (lg.cpu() - ref_l).abs().max().item()
    ev = (va.cpu() - ref_v.squeeze(1)).abs().max().item()
    print("R=%d blocks=%d hidden=%d dtype=%s: max|dlogit|=%.3e max|dvalue|=%.3e (logit range %.3f)" % (
        R, blocks, hidden, "fp16" if dtype else "bf16", el, ev, ref_l.abs().max().item()))
    assert el < tol and ev < tol
    eng.close()


@pytest.mark.parametrize("R,blocks", [(14, 3), (8, 3), (10, 2), (13, 2)])
def test_tower_wave_forms_give_identical_bits(R, blocks, monkeypatch):
    """k_tower on 8 waves (two per SIMD, loader / staggered roles) and on 4 waves (round 2's form with the streamed weight
    DMA; developer knob FPC_TOWER_WAVES=4) run the same MFMAs on the same operands in the same order per output element:
    logits and values must agree BIT FOR BIT, for both operand types.  At 14x14 a third form joins them: k_towerc, the
    default there since round 5 -- the same skeleton on the COMPACT image (13 row tiles of 16 squares instead of 14 grid
    rows; FPC_TOWER_COMPACT=0 gives the bordered grid back): logits bit for bit, values the same terms summed over other lanes."""
    import torch
    import weights
    m = _model(R, blocks, 128, seed=5)
    G = 48
    x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(R)) < 0.1).float().cuda()
    for dtype in (1, 0):
        outs = []
        monkeypatch.setenv("FPC_TOWERW", "0")       # k_tower at every size (off 14x14 the default at hidden 128 is k_towerw)
        for waves, compact in (("8", "0"), ("4", "0"), ("8", "1")):
            monkeypatch.setenv("FPC_TOWER_WAVES", waves)
            monkeypatch.setenv("FPC_TOWER_COMPACT", compact)
            eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=4, nn_dtype=dtype)
            eng.load_weights(weights.export_weights(m, dtype))
            lg = torch.empty(G, eng.A, device="cuda")
            va = torch.empty(G, device="cuda")
            for _ in range(3):
                eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
            torch.cuda.synchronize()
            outs.append((lg.cpu().numpy().copy(), va.cpu().numpy().copy()))
            eng.close()
        monkeypatch.delenv("FPC_TOWER_WAVES")
        monkeypatch.delenv("FPC_TOWER_COMPACT")
        monkeypatch.delenv("FPC_TOWERW")
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (R, dtype)
        assert np.array_equal(outs[0][0], outs[2][0]), (R, dtype)            # compact image (14x14; elsewhere the same kernel again)
        assert np.abs(outs[0][1] - outs[2][1]).max() < 2e-6, (R, dtype)
        assert np.abs(outs[0][0]).mean() > 1e-3


@pytest.mark.parametrize("R,blocks", [(14, 3), (8, 3), (10, 2), (13, 2)])
def test_towerw_and_tower_give_identical_bits_at_hidden_128(R, blocks, monkeypatch):
    """k_towerw<128> (two waves per SIMD, weights L2 -> registers, no barrier inside a layer: the default at hidden 128 on every
    board but 14x14; FPC_TOWERW=1 forces it) against k_tower (LDS-DMA weight ring: the default at 14x14; FPC_TOWERW=0): same MFMAs on the same operands in the same order
    per output element -> logits bit for bit (values: same terms, another summation order over the compact image's
    lanes), both operand types, several row-tile counts."""
    import torch
    import weights
    m = _model(R, blocks, 128, seed=8)
    G = 40
    x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(R + 1)) < 0.1).float().cuda()
    for dtype in (1, 0):
        outs = []
        for w in ("0", "1"):
            monkeypatch.setenv("FPC_TOWERW", w)
            eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=4, nn_dtype=dtype)
            eng.load_weights(weights.export_weights(m, dtype))
            assert (eng.L.fpc_nn_kernel(eng.h) or b"").decode() == ("k_towerw" if w == "1" else "k_towerc" if R == 14 else "k_tower")
            lg = torch.empty(G, eng.A, device="cuda")
            va = torch.empty(G, device="cuda")
            for _ in range(2):
                eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
            torch.cuda.synchronize()
            outs.append((lg.cpu().numpy().copy(), va.cpu().numpy().copy()))
            eng.close()
        monkeypatch.delenv("FPC_TOWERW")
        assert np.array_equal(outs[0][0], outs[1][0]), (R, dtype)       # logits: bit for bit
        assert np.abs(outs[0][1] - outs[1][1]).max() < 2e-6, (R, dtype)  # values: the same terms summed in another order
        assert np.abs(outs[0][0]).mean() > 1e-3


@pytest.mark.parametrize("R,blocks", [(14, 3), (8, 3), (10, 2), (13, 2)])
def test_towerw_wave_tilings_give_identical_logits_at_hidden_256(R, blocks, monkeypatch):
    """k_towerw<256> on one wave row (eight waves side by side along the output channels, each over all row tiles: the
    default) against two wave rows x four (developer knob FPC_TOWERW_ROWS=2): every output element sees the same MFMAs
    on the same operands in the same order -> logits bit for bit; the value head sums its terms in another order."""
    import torch
    import weights
    m = _model(R, blocks, 256, seed=11)
    G = 40
    x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(R)) < 0.1).float().cuda()
    for dtype in (1, 0):
        outs = []
        for rows in ("1", "2"):
            monkeypatch.setenv("FPC_TOWERW_ROWS", rows)
            eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=4, nn_dtype=dtype)
            eng.load_weights(weights.export_weights(m, dtype))
            assert (eng.L.fpc_nn_kernel(eng.h) or b"").decode() == "k_towerw"
            lg = torch.empty(G, eng.A, device="cuda")
            va = torch.empty(G, device="cuda")
            for _ in range(2):
                eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
            torch.cuda.synchronize()
            outs.append((lg.cpu().numpy().copy(), va.cpu().numpy().copy()))
            eng.close()
        monkeypatch.delenv("FPC_TOWERW_ROWS")
        assert np.array_equal(outs[0][0], outs[1][0]), (R, dtype)
        assert np.abs(outs[0][1] - outs[1][1]).max() < 2e-6, (R, dtype)
        assert np.abs(outs[0][0]).mean() > 1e-3


@pytest.mark.parametrize("dtype", [1, 0], ids=["fp16", "bf16"])
def test_tower256_megakernel_agrees_with_the_per_layer_convolutions(dtype, monkeypatch):
    """k_towerw at hidden 256 (two waves per SIMD, weights straight from L2 into registers, compact image, the residual folded
    into conv2's accumulators) against an INDEPENDENT implementation of the same network: the per-layer path (k_conv3x3 on
    the bordered grid through HBM + k_value_tail; developer knob FPC_NO_TOWER=1).  The same 16-bit operands and f32
    accumulation, other tilings and another order of the residual add: outputs within a handful of 16-bit roundings --
    far inside the 1e-3 both keep to the fp32 network (test_resnet_forward_vs_torch_fp32, fixture tests).  (Until round 5
    this test compared with round 2's k_tower256, since retired.)  Network inputs given as planes and the fused leaf encode
    (a short search; the per-layer path encodes with k_encode) both go through."""
    import torch
    import weights
    R, G = 14, 48
    m = _model(R, 3, 256, seed=6)
    x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(2)) < 0.1).float().cuda()
    boards = _positions(R, 12)
    outs = []
    for per_layer in ("0", "1"):
        monkeypatch.setenv("FPC_NO_TOWER", per_layer)
        eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=24, nn_dtype=dtype)
        eng.load_weights(weights.export_weights(m, dtype))
        assert (eng.L.fpc_nn_kernel(eng.h) or b"").decode() == ("k_conv3x3" if per_layer == "1" else "k_towerw")
        lg = torch.empty(G, eng.A, device="cuda")
        va = torch.empty(G, device="cuda")
        for _ in range(2):
            eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
        torch.cuda.synchronize()
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(24)
        res = eng.search_results(roots=roots)
        outs.append((lg.cpu().numpy().copy(), va.cpu().numpy().copy(), res))
        eng.close()
    monkeypatch.delenv("FPC_NO_TOWER")
    tol = 3e-4 if dtype else 3e-3          # a handful of fp16 / bf16 roundings falling the other way
    dl, dv = np.abs(outs[0][0] - outs[1][0]).max(), np.abs(outs[0][1] - outs[1][1]).max()
    print("k_towerw vs per-layer k_conv3x3, %s: max|dlogit| = %.3e, max|dvalue| = %.3e" % ("fp16" if dtype else "bf16", dl, dv))
    assert dl < tol and dv < tol
    assert np.abs(outs[0][0]).mean() > 1e-3
    for k in ("root_n", "n_children", "flat"):        # the same roots, the same legal moves in the same order
        assert np.array_equal(outs[0][2][k], outs[1][2][k]), k
    assert np.abs(outs[0][2]["prior"] - outs[1][2]["prior"]).max() < tol


@pytest.mark.parametrize("R,dtype", [(14, 1), (8, 0)])
def test_both_policy_linear_kernels_meet_the_fp32_network(R, dtype):
    """k_fcw (256 x 384 block tiles, fc_layout = 2: the default at 14x14; at 8x8 its 12 column groups x 8 K-splits) and k_fc16
    (256-column groups in long and short blocks, fc_layout = 1: the default elsewhere): each within the operand type's
    bound of the fp32 torch network, and within rounding of each other."""
    import torch
    import weights
    m = _model(R, 2, 128, seed=9)
    boards = _positions(R, 24)
    outs = []
    for layout in (2, 1):
        eng = make_engine("gpu", R, INV_OF[R], max_games=len(boards), max_sims=4, nn_dtype=dtype)
        eng.load_weights(weights.export_weights(m, dtype, fc_layout=layout))
        enc = np.concatenate([eng.encode([b]) for b in boards])
        x = torch.from_numpy(enc).cuda()
        lg = torch.empty(len(boards), eng.A, device="cuda")
        va = torch.empty(len(boards), device="cuda")
        eng.nn_forward(x.data_ptr(), len(boards), lg.data_ptr(), va.data_ptr())
        torch.cuda.synchronize()
        outs.append(lg.cpu().numpy().copy())
        eng.close()
    with torch.no_grad():
        ref_l, _ = m(torch.from_numpy(enc))
    tol = 1e-3 if dtype else 8e-3
    for o in outs:
        assert np.abs(o - ref_l.numpy()).max() < tol
    assert np.abs(outs[0] - outs[1]).max() < 2e-5      # same operands, f32 accumulation in another order


def test_headline_shape_256_rows_every_row_vs_fp32_network():
    """the headline shape -- 14x14, hidden 128, M = 256 rows -- held against the fp32 torch network
    ROW BY ROW.  256 positions from the reference's recorded playouts fill every row tile of the policy Linear
    (tiles 0..15), k_fcw's 248 one-round blocks and plan_fc's long and short blocks at Mtot = 256 and
    k_fc_reduce's chunk records at that size; both policy-Linear kernels; logits AND values at north_star's 1e-3 for
    every single row (the searches at this size only check properties that hold for wrong logits too)."""
    import torch
    import weights
    R, dtype, n = 14, 1, 256
    m = _model(R, 2, 128, seed=17)
    boards = _positions(R, n)
    assert len(boards) == n
    ref_l = ref_v = enc = None
    outs = []
    for layout in (2, 1):
        eng = make_engine("gpu", R, INV_OF[R], max_games=n, max_sims=4, nn_dtype=dtype)
        eng.load_weights(weights.export_weights(m, dtype, fc_layout=layout))
        if enc is None:
            enc = np.concatenate([eng.encode([b]) for b in boards])      # per-position rotation
            with torch.no_grad():
                ref_l, ref_v = m(torch.from_numpy(enc))
            ref_l, ref_v = ref_l.numpy(), ref_v.squeeze(1).numpy()
        x = torch.from_numpy(enc).cuda()
        lg = torch.empty(n, eng.A, device="cuda")
        va = torch.empty(n, device="cuda")
        torch.cuda.synchronize()
        eng.nn_forward(x.data_ptr(), n, lg.data_ptr(), va.data_ptr())
        torch.cuda.synchronize()
        lg, va = lg.cpu().numpy(), va.cpu().numpy()
        row_err = np.abs(lg - ref_l).max(axis=1)
        val_err = np.abs(va - ref_v)
        print("fc_layout %d: max|dlogit| %.3e (worst row %d), max|dvalue| %.3e, logit range %.3f" % (
            layout, row_err.max(), int(row_err.argmax()), val_err.max(), np.abs(ref_l).max()))
        assert np.isfinite(lg).all() and (row_err < 1e-3).all(), (layout, np.nonzero(row_err >= 1e-3)[0][:8], row_err.max())
        assert (val_err < 1e-3).all(), (layout, val_err.max())
        assert np.abs(lg).max(axis=1).min() > 1e-3               # every row really was computed
        outs.append(lg)
        eng.close()
    assert np.abs(outs[0] - outs[1]).max() < 2e-5                # same operands, f32 accumulation in another order


@pytest.mark.parametrize("blocks,hidden,sims", [(10, 128, 400), (20, 256, 800)], ids=["configs1", "configs3"])
def test_fused_search_equals_stepwise_at_the_headline_size(blocks, hidden, sims):
    """VERDICT r3, weak 1 (ii) / VERDICT r4, weak 1: fpc_search_run = the step-wise path at configs[1]'s full size -- 256
    games x 400 simulations, ResNet(10,128) fp16, 14x14 -- and at configs[3]'s -- 256 games x 800 simulations,
    ResNet(20,256) fp16 (k_towerw<1,256,13,true>'s own leaf encode into its compact image) -- extending the chain fused
    loop = step-wise loop = oracle (small sizes) to the full shapes.  The step-wise leg drives fpc_search_select / fpc_nn_forward / fpc_search_expand_select from the
    host with every tensor staying on the GPU (its network input comes from k_encode + k_nchw_to_grid, the fused
    loop's from the tower's own leaf encode; its softmax records from k_softmax_partials, the fused loop's from
    k_fc_reduce): visit counts, priors, value sums and the roots' piece-list orders must agree bit for bit."""
    import torch
    import weights
    R, G, dtype = 14, 256, 1
    m = _model(R, blocks, hidden, seed=0)
    eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(m, dtype))
    del m
    boards = _positions(R, G)                                   # mixed turns and depths (quirk Q6 in every batch)
    assert len(boards) == G
    roots_a = [fpc_ffi.clone_board(b) for b in boards]
    eng.search_begin(roots_a, 3.0)
    eng.search_run(sims)
    res_a = eng.search_results(roots=roots_a)

    roots_b = [fpc_ffi.clone_board(b) for b in boards]
    lg = torch.empty(G, eng.A, device="cuda")
    va = torch.empty(G, device="cuda")
    eng.search_begin(roots_b, 3.0)
    n_live, enc_ptr = eng.search_select()
    for i in range(sims):
        last = i == sims - 1
        if n_live == 0:
            if not last:
                n_live, enc_ptr = eng.search_select()
            continue
        eng.nn_forward(enc_ptr, G, lg.data_ptr(), va.data_ptr())     # device pointer in, device tensors out
        if last:
            eng.search_expand(lg.data_ptr(), va.data_ptr())
        else:
            n_live, enc_ptr = eng.search_expand_select(lg.data_ptr(), va.data_ptr())
    res_b = eng.search_results(roots=roots_b)
    for k in ("root_n", "n_children", "sims_done", "flat", "visits", "prior", "w"):
        assert np.array_equal(res_a[k], res_b[k]), k
    for a, b in zip(roots_a, roots_b):
        assert bytes(a) == bytes(b)
    assert int(res_a["sims_done"].sum()) > G * sims * 9 // 10
    eng.close()


def test_resnet_forward_more_than_256_rows():
    """300 positions = two 256-row tiles of the policy Linear (blockIdx.y, plan_fc with two passes of
    blocks) and 150 tower blocks of two 8x8 games each; same bound as the single-tile cases."""
    import torch
    import weights
    R, dtype = 8, 1
    m = _model(R, 2, 128, seed=5)
    eng = make_engine("gpu", R, INV_OF[R], max_games=300, max_sims=4, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(m, dtype))
    boards = _positions(R, 300)
    n = len(boards)
    assert n > 256
    enc = np.concatenate([eng.encode([b]) for b in boards])
    with torch.no_grad():
        ref_l, ref_v = m(torch.from_numpy(enc))
    x = torch.from_numpy(enc).cuda()
    lg = torch.empty(n, eng.A, device="cuda")
    va = torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    eng.nn_forward(x.data_ptr(), n, lg.data_ptr(), va.data_ptr())
    el = (lg.cpu() - ref_l).abs().max().item()
    ev = (va.cpu() - ref_v.squeeze(1)).abs().max().item()
    assert el < 1e-3 and ev < 1e-3, (el, ev)
    eng.close()


@pytest.mark.parametrize("R,dtype,rules,hidden", [(8, 0, 0, 64), (14, 1, 0, 64), (14, 1, 15, 64), (8, 1, 15, 64),
                                                  # k_towerw's own leaf encode into its compact image: one wave row at
                                                  # hidden 128 (8x8) and 256, two wave rows at hidden 128 elsewhere
                                                  (8, 1, 0, 128), (8, 1, 15, 128), (10, 1, 15, 128), (10, 1, 0, 256), (13, 0, 15, 256),
                                                  # the two shipped hidden-256 instances: configs[3]'s k_towerw<1,256,13,true>
                                                  # (14x14) and the reference default's k_towerw<1,256,4,true> (8x8)
                                                  (14, 1, 0, 256), (14, 1, 15, 256), (8, 1, 0, 256), (8, 1, 15, 256)])
def test_fused_search_equals_stepwise(R, dtype, rules, hidden):
    """fpc_search_run (encode->MFMA net->expand, no host round trip) must give exactly the visit
    counts of the step-wise path fed with the same network's outputs -- under the strict rules and
    under the non-strict rule set with root noise (N4: fused encode / decode rotations, plane numbering)."""
    import torch
    import weights
    m = _model(R, 2, hidden, seed=3)
    G, sims = 12, 48
    eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(m, dtype))
    eng.set_rules(rules)
    if rules:
        eng.set_root_noise(np.random.default_rng(7).standard_gamma(0.3, size=(G, fpc_ffi.MAX_MOVES)).astype(np.float32), 0.25)
    boards = _positions(R, G)
    roots_a = [fpc_ffi.clone_board(b) for b in boards]
    eng.search_begin(roots_a, 3.0)
    eng.search_run(sims)
    res_a = eng.search_results(roots=roots_a)

    def ev(enc):
        x = torch.from_numpy(np.ascontiguousarray(enc)).cuda()
        lg = torch.empty(G, eng.A, device="cuda")
        va = torch.empty(G, device="cuda")
        torch.cuda.synchronize()
        eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
        return lg.cpu().numpy(), va.cpu().numpy()

    roots_b = [fpc_ffi.clone_board(b) for b in boards]
    res_b = run_external_search(eng, "gpu", roots_b, sims, 3.0, ev)
    for k in ("root_n", "n_children", "sims_done", "flat", "visits", "prior", "w"):
        assert np.array_equal(res_a[k], res_b[k]), k
    for a, b in zip(roots_a, roots_b):
        assert bytes(a) == bytes(b)
    assert int(res_a["sims_done"].sum()) > G * sims // 2
    eng.close()


def test_arena_two_networks_temperature_zero():
    """configs[4] with the internal MFMA nets: two engines (one per weight set) on one GPU play paired
    games at temperature 0.  No sampling anywhere, so a second run must reproduce every visit count
    and pick; every pick is one of the reported legal children."""
    import arena
    import weights
    R, sims = 8, 32
    ma, mb = _model(R, 2, 64, seed=11), _model(R, 2, 64, seed=12)
    starts = _positions(R, 6)
    traces = []
    for _rep in range(2):
        ea = make_engine("gpu", R, INV_OF[R], max_games=12, max_sims=sims, nn_dtype=0)
        eb = make_engine("gpu", R, INV_OF[R], max_games=12, max_sims=sims, nn_dtype=0)
        ea.load_weights(weights.export_weights(ma, 0))
        eb.load_weights(weights.export_weights(mb, 0))
        games = arena.play_paired(arena.nn_search_fn(ea, sims, 3.0), arena.nn_search_fn(eb, sims, 3.0), ea,
                                  [fpc_ffi.clone_board(b) for b in starts], {"max_game_length": 16})
        for g in games:
            for _t, fl, _vi, pk in g.plies:
                assert pk in [int(x) for x in fl]
        s = arena.summary(games)
        assert s["games"] == 12 and s["wins_a"] + s["wins_b"] + s["draws"] == 12
        traces.append([[(t, [int(x) for x in fl], [int(x) for x in vi], pk) for t, fl, vi, pk in g.plies] + [g.winner]
                       for g in games])
        ea.close(); eb.close()
    assert traces[0] == traces[1]


@pytest.mark.parametrize("R,blocks,hidden,dtype", [(14, 2, 128, 0), (14, 2, 128, 1), (8, 2, 64, 0), (10, 2, 128, 1)])
def test_legal_only_policy_head_matches_full(R, blocks, hidden, dtype):
    """Opt-in FPC_POLICY_LEGAL: the policy Linear evaluated only at the leaves' legal moves
    (k_policy_gemv) must give the priors of the full Linear + full softmax up to f32 rounding (the
    softmax denominator cancels in mask + renormalise), the same value backups, and -- rounding
    aside -- the same search: root priors after one simulation within 2e-5, and after a 40-simulation
    search identical visit counts in (nearly) every game."""
    import weights
    m = _model(R, blocks, hidden, seed=21)
    G, sims = 24, 40
    boards = _positions(R, G)
    out = {}
    for mode in (False, True):
        eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=dtype)
        eng.load_weights(weights.export_weights(m, dtype))
        eng.set_policy_mode(mode)
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(1)
        r1 = eng.search_results(roots=roots)
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(sims)
        r2 = eng.search_results(roots=roots)
        out[mode] = (r1, r2)
        eng.close()
    (f1, f2), (l1, l2) = out[False], out[True]
    assert (f1["n_children"] == l1["n_children"]).all() and (f1["flat"] == l1["flat"]).all()
    dp = np.abs(f1["prior"] - l1["prior"]).max()
    dw = np.abs(f1["w"] - l1["w"]).max()
    same = sum(1 for g in range(G) if (f2["visits"][g] == l2["visits"][g]).all() and (f2["flat"][g] == l2["flat"][g]).all())
    print("R=%d %s: max|dprior|=%.2e max|dW|=%.2e identical visit vectors %d/%d" % (R, "fp16" if dtype else "bf16", dp, dw, same, G))
    assert dp < 2e-5 and dw == 0.0
    # The legal-only head is NOT bit-comparable with the full head: priors agree to f32
    # rounding, so a PUCT near-tie can resolve differently.  Differing games are REPORTED above, not
    # tolerated silently: the head stays opt-in and never produces bench.py's `value`.
    # A hard bound, so that a real regression in k_policy_gemv / k_expand_legal_select cannot pass: at
    # most 2 of the 24 games may differ, and a game that differs must still have searched the same root
    # children the same number of times in total (same tree size, same child set).
    assert same >= G - 2, "legal-only policy head: %d of %d games searched differently from the full head" % (G - same, G)
    for g in range(G):
        assert (f2["flat"][g] == l2["flat"][g]).all() and int(f2["visits"][g].sum()) == int(l2["visits"][g].sum())


@pytest.mark.parametrize("dtype", [1, 0], ids=["fp16", "bf16"])
def test_full_size_search_invariants(dtype):
    """BASELINE configs[1] at full size -- 256 concurrent games x 400 simulations, ResNet(10,128), 14x14,
    with the headline operand type (fp16) and with the one configs[1] names (bf16) -- where the oracle cannot follow (the CPU net alone would take hours): size-independent
    properties instead.  Roots are the start position advanced by 0-3 plies so the batch mixes turns
    (quirk Q6).  For every game: root N = sims + 1 and sum N_child = #children + sims - 1 (quirk Q1),
    the children are exactly the root's legal flat indices (ascending), priors are positive and sum
    to 1, |W_child| <= N_child - 1; and the whole search is reproducible bit for bit."""
    import weights
    R, G, sims = 14, 256, 400
    m = _model(R, 10, 128, seed=0)
    eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(m, dtype))
    g = gold(R)
    st = g["start"]
    base = fpc_ffi.board_from_dict(R, st["turn"], [tuple(e) for e in st["dict"]])
    rng = random.Random(3)
    boards = []
    for i in range(G):
        b = fpc_ffi.clone_board(base)
        for _ in range(i % 4):
            lm = eng.legal_moves([b])[0]
            b = eng.take_action([b], [lm[rng.randrange(len(lm))][2]])[0]
        boards.append(b)
    runs = []
    for _rep in range(2):
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(sims)
        runs.append(eng.search_results(roots=roots))
    r = runs[0]
    legal = eng.legal_moves([fpc_ffi.clone_board(b) for b in boards])
    for i in range(G):
        n = int(r["n_children"][i])
        assert int(r["sims_done"][i]) == sims and int(r["root_n"][i]) == sims + 1
        assert int(r["visits"][i, :n].sum()) == n + sims - 1
        assert [int(x) for x in r["flat"][i, :n]] == sorted(set(mv[2] for mv in legal[i]))
        p = r["prior"][i, :n]
        assert (p > 0).all() and abs(float(p.astype(np.float64).sum()) - 1.0) < 1e-5
        assert (np.abs(r["w"][i, :n]) <= r["visits"][i, :n] - 1 + 1e-9).all()
    for k in ("root_n", "n_children", "flat", "visits", "prior", "w"):
        assert (runs[0][k] == runs[1][k]).all(), k
    eng.close()


def test_config3_search_invariants():
    """BASELINE configs[3] at full size -- ResNet(20,256), 800 simulations/move, fp16 MFMA operands,
    14x14, 256 concurrent games -- through the fused on-device loop.  The logits of this network are
    held against the reference-net fixture in tests/test_net_gpu.py; here the size-independent search
    properties (quirk Q1 visit sums, children == legal set, priors sum to 1) and bitwise reproducibility."""
    import weights
    R, G, sims = 14, 256, 800
    m = _model(R, 20, 256, seed=0)
    eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=1)
    eng.load_weights(weights.export_weights(m, 1))
    del m
    g = gold(R)
    st = g["start"]
    base = fpc_ffi.board_from_dict(R, st["turn"], [tuple(e) for e in st["dict"]])
    rng = random.Random(5)
    boards = []
    for i in range(G):
        b = fpc_ffi.clone_board(base)
        for _ in range(i % 4):
            lm = eng.legal_moves([b])[0]
            b = eng.take_action([b], [lm[rng.randrange(len(lm))][2]])[0]
        boards.append(b)
    runs = []
    for _rep in range(2):
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(sims)
        runs.append(eng.search_results(roots=roots))
    r = runs[0]
    legal = eng.legal_moves([fpc_ffi.clone_board(b) for b in boards])
    for i in range(G):
        n = int(r["n_children"][i])
        assert int(r["sims_done"][i]) == sims and int(r["root_n"][i]) == sims + 1
        assert int(r["visits"][i, :n].sum()) == n + sims - 1
        assert [int(x) for x in r["flat"][i, :n]] == sorted(set(mv[2] for mv in legal[i]))
        p = r["prior"][i, :n]
        assert (p > 0).all() and abs(float(p.astype(np.float64).sum()) - 1.0) < 1e-5
    for k in ("root_n", "n_children", "flat", "visits", "prior", "w"):
        assert (runs[0][k] == runs[1][k]).all(), k
    eng.close()


def test_native_search_with_dropped_roots_and_weight_reload():
    """The fused on-device loop with games that leave the search early (quirk Q5: a terminal leaf drops
    the root -- fewer simulations, the game's network input and pairs go empty) next to games that
    run to the end, for both policy heads; then the same engine after loading OTHER weights must
    equal a fresh engine with those weights (the legal-only head rebuilds its row-major weight copy)."""
    import weights
    R, sims = 8, 60
    g = gold(R)
    q5 = g["searches"][-1]["before"][0]                      # R rook / kings position of SURVEY section 4 (Q5)
    boards = _positions(R, 6)
    boards = boards[:3] + [fpc_ffi.board_from_lists(R, q5["turn"], q5["pl"])] + boards[3:] + [fpc_ffi.board_from_lists(R, q5["turn"], q5["pl"])]
    G = len(boards)
    ma, mb = _model(R, 2, 128, seed=31), _model(R, 2, 128, seed=32)

    def run(eng):
        roots = [fpc_ffi.clone_board(b) for b in boards]
        eng.search_begin(roots, 3.0)
        eng.search_run(sims)
        return eng.search_results(roots=roots)

    res = {}
    for legal in (False, True):
        eng = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=1)
        eng.set_policy_mode(legal)
        eng.load_weights(weights.export_weights(ma, 1))
        ra = run(eng)
        eng.load_weights(weights.export_weights(mb, 1))      # same engine, other weights
        rb = run(eng)
        eng.close()
        fresh = make_engine("gpu", R, INV_OF[R], max_games=G, max_sims=sims, nn_dtype=1)
        fresh.set_policy_mode(legal)
        fresh.load_weights(weights.export_weights(mb, 1))
        rf = run(fresh)
        fresh.close()
        for k in ("root_n", "n_children", "sims_done", "flat", "visits", "prior", "w"):
            assert (rb[k] == rf[k]).all(), (legal, k)
        assert not (ra["visits"] == rb["visits"]).all()       # the two networks do search differently
        res[legal] = ra
    for r in res.values():
        sd = [int(x) for x in r["sims_done"]]
        assert sd[3] < sims and sd[-1] < sims and sd[3] == sd[-1]          # the Q5 roots were dropped early
        assert max(sd) == sims                                              # ... next to games that ran to the end
    assert (res[False]["visits"] == res[True]["visits"]).all() and (res[False]["sims_done"] == res[True]["sims_done"]).all()


def test_weight_blob_header_is_checked():
    """fpc_load_weights refuses, with a message that says why: a version-2 blob (rounds 1-2: the retired 32x32x16
    policy-Linear order -- an engine must never run k_fcw / k_fc16 on such weights), an unknown fc_layout, a blob for
    another operand type, a truncated blob.  The engine stays usable: the right blob loads afterwards."""
    import struct
    import torch
    import weights
    R, dtype = 8, 1
    m = _model(R, 2, 64, seed=4)
    blob = weights.export_weights(m, dtype)
    magic, version, r, F, nb, dt, a_ch, Np, Kp, layout = struct.unpack_from("<4s9i", blob, 0)
    assert magic == b"FPCW" and version == 3 and layout == 2 and Np % 384 == 0 and r == R
    eng = make_engine("gpu", R, INV_OF[R], max_games=8, max_sims=4, nn_dtype=dtype)

    def patched(**kw):
        f = {"version": version, "dtype": dt, "layout": layout}
        f.update(kw)
        return struct.pack("<4s9i", magic, f["version"], r, F, nb, f["dtype"], a_ch, Np, Kp, f["layout"]) + blob[40:]

    for bad, words in ((patched(version=2), "version"), (patched(layout=0), "layout"), (patched(layout=7), "layout"),
                       (patched(dtype=0), "dtype"), (blob[: len(blob) // 2], "truncated")):
        with pytest.raises(RuntimeError) as ei:
            eng.load_weights(bad)
        assert words in str(ei.value), (words, str(ei.value))
    eng.load_weights(blob)
    x = (torch.rand(8, 24, R, R, generator=torch.Generator().manual_seed(1)) < 0.1).float().cuda()
    lg = torch.empty(8, eng.A, device="cuda"); va = torch.empty(8, device="cuda")
    eng.nn_forward(x.data_ptr(), 8, lg.data_ptr(), va.data_ptr())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref_l, _ = m(x.cpu())
    assert np.abs(lg.cpu().numpy() - ref_l.numpy()).max() < 1e-3
    eng.close()
