"""TEST INFRASTRUCTURE ONLY — CPU restatement of the Chen001117/mappo hot path.

Nothing under ``mappo_amd/`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Parity status: PINNED against outputs of the reference itself.  The reference ships no
tests or golden vectors, so ``tests/golden/generate_golden.py`` imports the
reference's hot-path modules in the build container (torch 2.10 / numpy 2.2, see SURVEY.md
§8c) and commits their inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function of this package against them.
"""
