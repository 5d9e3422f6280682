"""A short, seeded session of tests/fuzz_parity.py: random tree shapes and numberings x random handle options x random
batches and entry points, bit for bit against the oracle.  (Longer sessions by hand: python tests/fuzz_parity.py 900.)"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,big", [(20261003, False), (7, True)])
def test_seeded_fuzz_session(seed, big, monkeypatch, tmp_path):
    import fuzz_parity
    monkeypatch.setenv("SUCHTREE_AMD_CACHE_DIR", str(tmp_path / "tune"))      # (records of the timed deep trees: not in ~/.cache)
    assert fuzz_parity.run(budget=25.0 if not big else 20.0, seed=seed, big=big)
