"""CPU oracle for the detect -> refine -> uplift hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``upliftingtabletennis_amd`` may import this
package: it is the checker for the HIP path, never the thing shipped or measured.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` import it.

Every function restates one piece of the reference (KieDani/UpliftingTableTennis)
in plain torch-CPU / numpy / scipy and cites the reference file:line it follows.

Pinning (how this oracle is itself checked):
  * ``tools/make_goldens.py`` imports the *reference's own Python modules* from
    ``/root/reference`` in the build container, runs them on seeded inputs and
    random (seeded) weights and stores inputs + outputs as ``tests/golden/*.npz``.
  * ``tests/test_oracle_golden.py`` checks every oracle function against those
    files, so the oracle == reference on torch 2.10 CPU / scipy 1.15.
  * Unpinned pieces (no reference counterpart importable here) say so in their
    own docstring: ``cv2.resize`` (OpenCV absent) -> parity unpinned.
"""
