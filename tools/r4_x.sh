python -m pytest tests/test_gpu_sharded_paths.py -x -q -m gpu -k "reviews_itself" 2>&1 | tail -8
python3 - <<'PY'
import re,sys
src=open('tests/test_gpu_sharded_paths.py').read()
m=re.search(r"_REVIEW_SCRIPT = r'''(.*?)'''", src, re.S)
for flag in (False, True):
    code="ROOT='.'\nFIRST_ON_NULL_STREAM=%s\n" % flag + m.group(1)
    import subprocess
    p=subprocess.run([sys.executable,'-c',code],capture_output=True,text=True)
    print(flag, [l for l in p.stdout.splitlines() if l.startswith('{')], p.stderr[-300:] if p.returncode else '')
PY
