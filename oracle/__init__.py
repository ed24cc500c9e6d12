"""CPU oracle for the StoRM reverse-SDE sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / timed baseline.  The
product path (``storm_amd``) never imports this package and fails loudly
when its HIP library is missing.

What it is: a plain PyTorch-CPU fp32 restatement of the reference algorithm
for the path ``ScoreModel.enhance -> get_pc_sampler -> predictor/corrector ->
OUVE SDE -> NCSN++ forward`` plus the STFT front/back end.  Every function
cites the reference file:line it follows (paths relative to /root/reference).

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against the reference itself:
``oracle/make_golden.py`` imports the reference in the build container
(stub recipe in ``oracle/ref_import.py``), runs it on seeded inputs, checks
this restatement against it and writes the input/output vectors to
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the
restatement against those committed vectors everywhere (no reference needed).
"""
