"""A short run of the randomised fused-vs-composed sweep (tests/fuzz_fused.py): ragged ray / sample counts, every dataset
preset, 2..16 levels, dense-only to fully hashed tables, single / dual field, pose gradients on and off, random subsets
of the outputs feeding the loss."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations(seed):
    from fuzz_fused import run
    failures, _ = run(8, seed, verbose=False)
    assert not failures, failures
