"""Pin the oracle (oracle/np_oracle.c) against the reference's own known-answer tests.

tests/golden/phpt_vectors.json holds the inputs / calls / --EXPECT-- text of the reference's
tests/math/002..044-*.phpt and tests/linalg/001-ndarray-matmul.phpt.  Each record is replayed
through the oracle and the print_r text is compared byte for byte (PHP prints fp32 values widened
to double with 14 significant digits, so a 1-ulp difference in fp32 changes the text).
"""
import pytest

from tests.phpt_replay import OracleBackend, load_vectors, replay

TESTS = load_vectors()


@pytest.mark.parametrize("test", TESTS, ids=[t["source"].split("/")[-1] for t in TESTS])
def test_oracle_matches_phpt_expect(test, oracle):
    got = replay(OracleBackend(), test)
    assert got.rstrip() == test["expect"].rstrip()
