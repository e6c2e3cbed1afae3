import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
print(json.dumps(bench.sdfnet_numbers(), indent=1))
