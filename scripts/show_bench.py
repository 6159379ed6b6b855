import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "img/s", round(d["value"], 1))
for k, v in d["roofline"]["families"].items():
    print("   ", k, round(v["ms_per_step"], 3), v["launches_per_step"], round(v["tflops"], 1))
for k, v in list(d["roofline"]["by_shape"].items())[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("      ", k, v["calls_per_step"], round(v["us_per_call"], 1), round(v["roofline_us"], 1))
