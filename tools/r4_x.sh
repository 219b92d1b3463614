python3 - <<'PY'
import re,sys,subprocess
src=open('tests/test_gpu_sharded_paths.py').read()
m=re.search(r"_REVIEW_SCRIPT = r'''(.*?)'''", src, re.S)
for flag in (False, True, True):
    code="ROOT='.'\nFIRST_ON_NULL_STREAM=%s\n" % flag + m.group(1)
    p=subprocess.run([sys.executable,'-c',code],capture_output=True,text=True)
    print(flag, [l for l in p.stdout.splitlines() if l.startswith('{')], p.stderr[-300:] if p.returncode else '')
PY
