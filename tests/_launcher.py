"""Child-process launcher for multi-process GPU tests.

pytest starts this helper BEFORE anything in the test process has touched the GPU (conftest.pytest_configure), so that
the torch.distributed.run jobs it starts later are children of a process that never initialised HIP: on the GPU pool a
process that has initialised the GPU must not exec another program.  Protocol: one JSON request per stdin line
{"argv": [...], "env": {...}, "timeout": seconds} -> one JSON reply per stdout line {"rc": int, "tail": str}."""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        try:
            p = subprocess.run(req["argv"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=req.get("timeout", 600), cwd=req.get("cwd") or None)
            rc, tail = p.returncode, p.stdout.decode(errors="replace")[-6000:]
        except subprocess.TimeoutExpired as e:
            rc, tail = 124, "timeout\n" + ((e.stdout or b"").decode(errors="replace")[-6000:])
        sys.stdout.write(json.dumps({"rc": rc, "tail": tail}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
