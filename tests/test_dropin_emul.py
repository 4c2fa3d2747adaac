import pytest

import dropin_cases as dc


@pytest.mark.parametrize("R", [8, 14])
def test_reference_style_usage(R):
    assert dc.case_reference_style_usage("emul", R)


def test_training_loop():
    assert dc.case_training_loop("emul") >= 1
