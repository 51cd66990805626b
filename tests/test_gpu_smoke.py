import pytest
import torch

pytestmark = pytest.mark.gpu


def test_smoke_step_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.smoke_impl import run_smoke
    run_smoke("cuda:0", verbose=False)
