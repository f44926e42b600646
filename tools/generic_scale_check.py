import subprocess, time, os, sys
sys.path.insert(0, os.getcwd())
from graphite_amd import synth
prob = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/lb1723.txt", prob)
for mode in ("engine",):
    for solver in ("pcg", "pcg-schur"):
        t = time.time()
        r = subprocess.run(["build/test_generic_bal", "/tmp/lb1723.txt", solver, "10", mode], capture_output=True, text=True, env=dict(os.environ, GR_VERBOSE="1"))
        print(mode, solver, "wall %.2fs" % (time.time() - t), "rc", r.returncode)
        print("\n".join(r.stdout.splitlines()[:3] + r.stdout.splitlines()[-5:-2]))
        print(r.stderr[-600:])
