"""The random call sequences of tests/test_gpu_fuzz.py run on the CPU oracle alone (two oracle engines side by side):
no GPU needed, structural invariants checked at the end of every trial.  Under an AddressSanitizer build of the oracle
(oracle/Makefile: liboracle_asan.so) this is the run that found the visible-list overflow."""
import pytest

import test_gpu_fuzz


@pytest.mark.parametrize("seed", range(1000, 1015))
def test_oracle_random_call_sequences(pkg, synth, oracle, seed):
    import __graft_entry__ as ge
    second = ge.load_oracle().open_oracle(pkg.CApi)
    test_gpu_fuzz._trial(pkg, synth, oracle, second, seed)
