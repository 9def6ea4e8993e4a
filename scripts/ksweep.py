import subprocess, json, sys
for k in (10, 20, 40, 80, 160, 320, 640):
    vals = []
    for rep in range(3):
        o = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", str(k), "--warmup", "20", "--no-cpu-baseline",
                            "--no-roofline", "--no-serving"], capture_output=True, text=True).stdout
        vals.append(json.loads(o.strip().splitlines()[-1])["ms_per_step"] * 1e3)
    print(k, ["%.1f" % v for v in vals], "total us: %.0f" % (min(vals) * k))
