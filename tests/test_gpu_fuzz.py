"""A short run of the randomised differential test (tests/fuzz_util.py; tools/dev/dev_fuzz.py runs it for as long as one likes)."""
import pytest
from fuzz_util import run_fuzz

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_traffic_and_settings_against_the_oracle(fx, oracle, seed):
    n, nbad = run_fuzz(fx, oracle, 20, seed, verbose=False)
    assert n > 2000 and nbad > 100, (n, nbad)
