"""oracle/ -- CPU restatement of the reference's CFFM hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*: a plain PyTorch-CPU (fp32/fp64) restatement of
GuoleiSun/VSS-CFFM's coarse-to-fine cross-frame attention path, written from the
numerical contract in SURVEY.md Appendix A with explicit index maps (no roll /
unfold / cat), each function citing the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the thing results are
compared with (or timed next to), never as the thing measured or shipped.
Nothing under ``vss_cffm_amd/`` imports this package; the product path raises
if the HIP library is missing instead of falling back to anything here.

Parity pin: the oracle is checked in ``tests/test_oracle_vs_reference.py``
against the reference's own modules imported from /root/reference (when that
tree is present, i.e. in the build container) and against the golden vectors in
``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` generated from those
same reference modules (always, also on the GPU box).  The reference's own test
suite holds no vectors for this path.
"""
