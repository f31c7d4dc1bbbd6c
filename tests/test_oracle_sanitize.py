"""The oracle (test infrastructure) itself under AddressSanitizer + UBSan on a synthetic
map: scans incl. out-of-map poses, two-agent env steps, GJK, ray cast, lap logic."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'orc_san')
    cmd = ['gcc', '-O1', '-g', '-std=c11', '-ffp-contract=off', '-fsanitize=address,undefined',
           '-fno-sanitize-recover=all', '-I', os.path.join(ROOT, 'oracle'), '-o', exe,
           os.path.join(ROOT, 'oracle', 'sanitize_main.c'), '-lm']
    subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, 'oracle'))
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith('ok '), r.stdout
