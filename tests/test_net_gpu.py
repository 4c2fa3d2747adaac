"""The HIP network and search on a real MI355X against fixtures recorded from the reference's OWN
net.py (oracle/gen_net_golden.py): north_star's float bar -- policy/value logits within 1e-3 of
the reference's fp32 PyTorch-CPU path -- on the configurations the benchmark runs, plus the
recorded-network and 800-simulation searches and the training batch."""
import numpy as np
import pytest

import net_cases as nc
from fpc_testlib import make_engine

pytestmark = pytest.mark.gpu

TOL = 1e-3          # BASELINE.json north_star: "policy/value logits within 1e-3"


def _forward_vs_fixture(R, blocks, hidden, dtype):
    import torch
    import weights
    fx = nc.load_net_fixture(R, blocks, hidden)
    model = nc.fixture_model(fx)                       # checksum-proven equal to the reference's net.py module
    boards = nc.fixture_boards(fx)
    n = len(boards)
    eng = make_engine("gpu", R, nc.INV_OF[R], max_games=n, max_sims=4, nn_dtype=dtype)
    eng.load_weights(weights.export_weights(model, dtype))
    del model
    enc = np.concatenate([eng.encode([b]) for b in boards])          # per-position rotation, as the fixture
    x = torch.from_numpy(enc).cuda()
    lg = torch.empty(n, eng.A, device="cuda")
    va = torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    eng.nn_forward(x.data_ptr(), n, lg.data_ptr(), va.data_ptr())
    lg = lg.cpu().numpy()
    el = float(np.abs(lg[:, fx["idx"]] - fx["logits"]).max())
    ev = float(np.abs(va.cpu().numpy() - fx["value"]).max())
    es = float(np.abs(lg.astype(np.float64).sum(axis=1) - fx["rowsum"]).max() / lg.shape[1])     # mean error over ALL A logits
    # EVERY column of four rows against the reference's net.py: folded into the bound the caller asserts
    ef = float(np.abs(lg[fx["full_rows"]] - fx["full_logits"]).max())
    assert fx["full_logits"].shape == (len(fx["full_rows"]), eng.A) and len(fx["full_rows"]) >= 4
    el = max(el, ef)
    print("reference-net fixture R=%d ResNet(%d,%d) %s: max|dlogit|=%.3e (all columns of %d rows: %.3e) max|dvalue|=%.3e mean-row-err=%.2e (|logit|max %.3f)" % (
        R, blocks, hidden, "fp16" if dtype else "bf16", el, len(fx["full_rows"]), ef, ev, es, float(fx["absmax"].max())))
    eng.close()
    return el, ev


@pytest.mark.parametrize("R,blocks,hidden", [(14, 10, 128), (14, 20, 256), (8, 10, 128), (8, 4, 64), (8, 15, 256)])
def test_logits_vs_reference_net_fixture_fp16(R, blocks, hidden):
    """The headline operand type (bench.py's `dtype`): fp16 MFMA operands, f32 accumulation.  configs[1]'s
    ResNet(10,128) and configs[3]'s ResNet(20,256) at 14x14, configs[0]'s ResNet(4,64) at 8x8.  The bound
    is north_star's 1e-3, not widened."""
    el, ev = _forward_vs_fixture(R, blocks, hidden, 1)
    assert el < TOL and ev < TOL, (el, ev)


BF16_MEASURED_BOUND = 8e-3


@pytest.mark.parametrize("R,blocks,hidden", [(14, 10, 128), (14, 20, 256), (8, 10, 128), (8, 4, 64), (8, 15, 256)])
def test_logits_vs_reference_net_fixture_bf16_reported(R, blocks, hidden):
    """bf16 MFMA operands do NOT meet north_star's 1e-3 on these networks: 8 significand bits on every
    weight and activation give max|dlogit| = 1.5e-3 .. 4.9e-3 against the reference's fp32 path (fp16, 11
    bits, same MFMA rate: 2e-4 .. 6e-4).  No parity claim is made for bf16 -- it is measured, printed
    and reported beside the fp16 headline in bench.py's JSON; this test only guards against regressions
    of that measured error (bound 8e-3) and documents that 1e-3 is out of reach."""
    el, ev = _forward_vs_fixture(R, blocks, hidden, 0)
    assert el < BF16_MEASURED_BOUND and ev < BF16_MEASURED_BOUND, (el, ev)
    assert el > TOL / 2, "bf16 unexpectedly close to the fp32 reference: re-evaluate the headline dtype"


@pytest.mark.parametrize("R", [8, 14])
def test_recorded_net_search(R):
    assert nc.case_recorded_net_search("gpu", R) >= 10


@pytest.mark.parametrize("R", [8, 14])
def test_search_800(R):
    assert nc.case_search_800("gpu", R) == 2


def test_train_batch_and_loss():
    assert nc.case_train_batch("gpu") == 16
