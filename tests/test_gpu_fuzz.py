"""A slice of tests/gpu_fuzz.py inside the GPU suite: seeded random problems (distribution, size,
boundaries, softening variant, walk variant, active list) through tree, walks, density and hydro
against the oracle.  `python tests/gpu_fuzz.py N` runs more seeds."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(1000, 1010))
def test_random_problem_matches_the_oracle(seed):
    import gpu_fuzz
    gpu_fuzz.one(seed)
