"""CPU side of the fixtures recorded from the reference's own net.py / mcts.py / fen_parser.py by
oracle/gen_net_golden.py: the oracle and the wavefront-emulator build of the tree kernels against
the recorded-network search and the 800-simulation searches, our net.py module against the
reference's logits, the start layouts, and the training batch + loss."""
import numpy as np
import pytest

import net_cases as nc


@pytest.mark.parametrize("R", [8, 14])
def test_recorded_net_search_oracle(R):
    assert nc.case_recorded_net_search("oracle", R) >= 10


@pytest.mark.parametrize("R", [8, 14])
def test_recorded_net_search_emul(R):
    assert nc.case_recorded_net_search("emul", R) >= 10


@pytest.mark.parametrize("R", [8, 14])
def test_search_800_oracle(R):
    assert nc.case_search_800("oracle", R) == 2


def test_search_800_emul():
    assert nc.case_search_800("emul", 8, kinds=("hash",)) == 1


def test_start_layouts():
    assert nc.case_start_layouts() == 5


def test_train_batch_and_loss():
    assert nc.case_train_batch("emul") == 16


@pytest.mark.parametrize("R,blocks,hidden", [(8, 4, 64), (8, 10, 128), (8, 15, 256)])
def test_module_equals_reference_net(R, blocks, hidden):
    """our net.py module, built under the fixture's seed, holds the reference net.py's weights
    (checksums) and reproduces its fp32 logits and values on the golden positions."""
    import torch
    fx = nc.load_net_fixture(R, blocks, hidden)
    model = nc.fixture_model(fx)
    from oracle import orc
    g = nc.gold(R)
    boards = [orc.board_from_lists(R, g["playouts"][int(a)][int(b)]["before"]["turn"], g["playouts"][int(a)][int(b)]["before"]["pl"]) for a, b in fx["pos"]]
    enc = np.concatenate([orc.encode([b], R) for b in boards])
    with torch.no_grad():
        lg, va = model(torch.from_numpy(enc))
    assert np.abs(lg.numpy()[:, fx["idx"]] - fx["logits"]).max() < 2e-6
    assert np.abs(lg.numpy()[fx["full_rows"]] - fx["full_logits"]).max() < 2e-6      # every column of four rows
    assert np.abs(va.squeeze(1).numpy() - fx["value"]).max() < 2e-6
    assert (lg.argmax(dim=1).numpy() == fx["argmax"]).all()
