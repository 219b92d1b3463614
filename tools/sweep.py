#!/usr/bin/env python3
"""One runner for the measurement recipes of this repo (tools/sweeps.json), run ON THE GPU BOX through gpurun:

    gpurun -- python3 tools/sweep.py sobel_ns4 [another_recipe ...]
    python3 tools/sweep.py --list

A recipe = what it measures, the commands (run in order, each under `timeout`; a failing or timed-out command stops the recipe -- no GPU
step is started after one was killed), environment for all of them, and the file under gpurun_out/sweeps/ that collects their output.
The figures DESIGN.md / NOTES.md quote name the recipe and the file under profiles/ the output was copied to (profiles/MANIFEST.md)."""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECIPES = json.load(open(os.path.join(ROOT, "tools", "sweeps.json")))
# round 2's sixty-seven one-off call scripts (tools/r2_call_<x>.sh), folded in as recipes r2_<x>: same commands, one runner
RECIPES.update(json.load(open(os.path.join(ROOT, "tools", "sweeps_r2.json"))))


def main():
    args = sys.argv[1:]
    if not args or args[0] == "--list":
        for name, r in RECIPES.items():
            print(f"{name:28s} {r['what']}")
        return 0
    outdir = os.path.join(ROOT, "gpurun_out", "sweeps")
    os.makedirs(outdir, exist_ok=True)
    for name in args:
        r = RECIPES[name]
        path = os.path.join(outdir, name + ".txt")
        env = dict(os.environ, **{k: str(v) for k, v in r.get("env", {}).items()})
        # the recipes turn GH_* tuning knobs: those exist in the probe build only (`make probes`); "lib": "release" in a recipe = the shipped library
        if r.get("lib", "probes") == "probes":
            env.setdefault("GRAYHIP_LIB", os.path.join(ROOT, "build_probes", "libgrayhip_probes.so"))
        with open(path, "w") as f:
            f.write(f"# {name}: {r['what']}\n")
            for cmd in r["cmds"]:
                f.write(f"$ {cmd}\n")
                f.flush()
                t0 = time.time()
                p = subprocess.run(["timeout", "-k", "10", str(r.get("timeout", 300)), "bash", "-o", "pipefail", "-c", cmd], cwd=ROOT, env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                keep = [ln for ln in p.stdout.splitlines() if "amdgpu.ids" not in ln]
                f.write("\n".join(keep) + f"\n# rc {p.returncode}, {time.time() - t0:.0f} s\n")
                f.flush()
                print(f"[{name}] {cmd[:100]} -> rc {p.returncode}", flush=True)
                if p.returncode != 0:
                    print("\n".join(keep[-15:]))
                    return p.returncode
        print(open(path).read()[-int(r.get("tail", 3000)):])
    return 0


if __name__ == "__main__":
    sys.exit(main())
