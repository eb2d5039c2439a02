"""The pieces of bench.py (the contract benchmark at the repo root): workloads and their synthetic inputs (workloads), the CPU baseline on the
oracle (cpu), rocprofv3 PMC passes / committed profiles / per-kernel accounting (pmc), the sampling and training directions (directions), the
self-launch of the ranks and the dry run (launcher), what is measured after the timed region (sweeps).  bench.py keeps the command line, the
log-prob timing loop and the assembly of the JSON line."""
