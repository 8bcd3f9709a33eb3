import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1))
    for k, v in d["other_configs"].items():
        print("   ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
