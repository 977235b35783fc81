"""Slices of tests/gpu_fuzz.py and tests/gpu_ddfuzz.py inside the GPU suite: seeded random problems (distribution, size,
boundaries, softening variant, walk variant, active list) through tree, walks, density and hydro
against the oracle.  `python tests/gpu_fuzz.py N` runs more seeds."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(1000, 1010))
def test_random_problem_matches_the_oracle(seed):
    import gpu_fuzz
    gpu_fuzz.one(seed)


@pytest.mark.parametrize("seed", [5000, 5003, 5011, 5027, 5036, 6017, 6101, 6204])
def test_random_problem_on_random_shards_matches_the_oracle(seed):
    """tests/gpu_ddfuzz.py: the same kind of random problem cut into 2..8 logical shards (some nearly
    empty), gravity + SPH + a shake with migration, against the oracle's single tree."""
    import gpu_ddfuzz
    gpu_ddfuzz.one(seed)
