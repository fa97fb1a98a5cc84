"""CPU oracle for the collector -> replay_buffer -> algo.update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``torchrl_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.

Every function restates, in this repo's own words, the algorithm of the
RchalYang/torchrl reference (cited as ``file:line`` relative to the reference
root).  Parity pin: ``tests/golden/*.npz`` were produced by importing the
reference itself in the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks every oracle function against them.
"""
