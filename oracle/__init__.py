"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference hot path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU baseline.
The product package ``salience_detr_amd`` never imports it.

Pinned against golden vectors generated from the imported upstream reference
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see DESIGN.md.
"""
