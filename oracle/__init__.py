"""ORACLE — test infrastructure only (never imported by the product package).

CPU restatements of LCR-Net's per-scan hot path used as the parity checker:
  * ``oracle.ops``        — ctypes front-end to ``oracle/_build/liblcr_oracle.so`` (C++ restatement of
                            grid subsampling + radius search) and, when present, ``oracle/_ref/libref_ops.so``
                            (the reference's own C++ compiled from /root/reference by ``oracle/Makefile``).
  * ``oracle.torch_ref``  — plain PyTorch fp32 restatement of the floating-point modules
                            (KPConv blocks, GroupNorm, NetVLAD, 3D-RoFormer attention, retrieval).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
"""
