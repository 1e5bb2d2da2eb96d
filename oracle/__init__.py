"""CPU oracle for the AD-YOLO hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Everything under ``oracle/`` is a CPU restatement (NumPy float64 for the feature
pipeline, PyTorch-CPU float32 for encoder / head / loss) of the reference
algorithm (sadPororo/AD-YOLO, ``/root/reference/src``).  Each function cites the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker /
baseline: the product package (``ad-yolo_amd``) never imports ``oracle`` and
fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * encoder / head / loss / label encoder / collate: pinned by golden vectors
    generated from the real reference in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * feature pipeline (STFT / log-mel / intensity vector): the arithmetic lives
    in librosa==0.8.1 which is absent from /root/reference and from this image:
    "parity unpinned" at that boundary; cross-checked against ``torch.stft`` and
    ``transformers.audio_utils.mel_filter_bank`` only.
"""
