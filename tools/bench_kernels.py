import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "gru" in k})
