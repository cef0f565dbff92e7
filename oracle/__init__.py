"""CPU oracle for the qpwcnet CostVolume + Warp hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``qpwcnet_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED (stated as the task contract requires): the reference
(yycho0108/qpwcnet) ships no stored golden vectors for this path, and it cannot
be executed in the authoring container because TensorFlow and
tensorflow-addons are not installed (plain ``ModuleNotFoundError``; there is no
network to install them).  What pins the restatement instead:

* the CostVolume arithmetic is fully specified in-tree
  (``qpwcnet/core/layers.py:72-100``) and restated here op for op;
* the reference's own invariant ``CostVolume == CostVolumeV2``
  (``qpwcnet/app/test/test_cvol_equal.py:25``) makes that source the spec for
  the tfa ``CorrelationCost`` variant as well; the tfa op (third-party,
  ``tensorflow_addons``, version unpinned) is ALSO restated on its own from its
  published algorithm in ``oracle/tfa_ref.py`` / ``oracle_correlation_cost`` (C), the
  two restatements are asserted equal at the reference's shapes, and the GPU
  ``CostVolumeV2`` is tested against that one;
* ``tf_warp`` (``qpwcnet/core/warp.py:63-153``) is in-tree and restated op for
  op; ``WarpV2`` follows the in-tree copy of tfa ``dense_image_warp``
  (``qpwcnet/core/warp.py:156-211``) plus the published algorithm of
  ``tfa.image.interpolate_bilinear`` (tensorflow-addons, version unpinned by the
  reference's ``setup.py:17-20``);
* analytic known answers derived from the reference's test scripts
  (``qpwcnet/app/optical_flow/test_warp.py:28-33``, ``qpwcnet/core/vis.py:22-32``)
  are checked in ``tests/test_oracle.py``.
"""
