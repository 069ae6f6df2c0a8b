"""CPU oracle for the MFCC front-end.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``mfcc_amd``) may
import, call or execute anything in this directory; the only permitted
callers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.

Parity pinned: ``oracle/gen_golden.py`` ran the reference's own functions
(``mfcc-self5.py`` / ``mfcc-self15.py`` / ``MLP_5hidlayer_40.py``) in the build
container and committed their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against every one of
those vectors.
"""
