"""CPU oracle for the YOLOv8-seg hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 ops + numpy) of the arithmetic
behind the reference's `self.model.predict(...)` call (measurement.py:208-210,
Utils/check_model.py:331-337) and of measurement.py's mask post-processing
(measurement.py:70-86,160-185,300-330).

PARITY UNPINNED: the arithmetic lives in the third-party `ultralytics` package
(unpinned, requirements.txt:13; plus torchvision.ops.nms and
opencv-contrib-python==4.11.0.86, requirements.txt:2).  None of them is
installed here, the reference ships no tests, golden vectors or weights
(.MISSING_LARGE_BLOBS), so this oracle restates the published Ultralytics 8.x
algorithm and is pinned only by (a) the published model.info() parameter counts
and (b) hand-computed fixtures under tests/golden/.  See DESIGN.md section "Oracle".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (vti_amd / libvti.so) never does.
"""
