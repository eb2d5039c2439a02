"""probe: the numpy oracle's aggregate rate on the host for several worker counts / chunk sizes (bench.cpu_baseline)"""
import os, sys
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print("cpus", os.cpu_count(), bench.cpu_model())
for workers, chunk in ((64, 4096), (128, 4096), (32, 4096), (128, 2048), (96, 2048), (64, 8192)):
    r = bench.cpu_baseline("c3", budget_s=4.0, workers=workers, chunk=chunk)
    print(workers, chunk, "%.3g evals/s  per worker %.3g" % (r["value"], r["per_core"]))
