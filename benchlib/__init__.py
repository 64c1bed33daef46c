"""The measurement legs of bench.py (repo root), split out so that bench.py itself is the timed region, the parity legs and the
CPU baseline.  Nothing here imports the CPU restatement: the legs that spot-check outputs receive it from bench.py as ctx["checker"].

  roofline.py  peaks, the requantisation ceiling measured in the run, replay of the committed counter passes, HIP-event timing
  compact.py   the ONE compact JSON line the driver parses (and the full record beside it)
  records.py   the other workloads and sub-records of the default line (speech, fc4096, run-time geometry, general conv / depthwise)
"""
