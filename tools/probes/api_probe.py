import sys, json
sys.path.insert(0, "/root/repo")
import bench
class A: pass
print(json.dumps(bench.api_path(A(), 128, "bilinear-diag", "min", 512), indent=1))
