"""Shader clock and socket power while the headline step runs (rocm-smi polled from a second process): the dense-MFMA peak the roofline
fraction is priced against (2.5 PFLOP/s) assumes the 2.4 GHz boost clock.  usage: python tools/probes/clock_sample.py [steps]"""
import json, os, subprocess, sys, time
steps = sys.argv[1] if len(sys.argv) > 1 else '3000'
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        pick = lambda sub: next((v for k, v in c.items() if sub in k.lower()), None)
        return pick('sclk clock speed'), pick('mclk clock speed'), pick('socket graphics package power') or pick('power (w)') or pick('average graphics'), pick('junction')
    except Exception as e:      # noqa
        return ('?', '?', '?', repr(e)[:60])


print('idle            sclk %s  mclk %s  power %s W  junction %s C' % sample(), flush=True)
p = subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--steps', steps, '--warmup', '10', '--no-cpu-baseline', '--no-roofline'],
                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
t0 = time.time()
while p.poll() is None:
    time.sleep(1.0)
    print('t = %5.1f s      sclk %s  mclk %s  power %s W  junction %s C' % ((time.time() - t0,) + sample()), flush=True)
line = [l for l in p.stdout.read().splitlines() if l.startswith('{')]
if line:
    d = json.loads(line[-1])
    print('bench: %.1f images/s, %.4f ms per step over %s steps' % (d['value'], d['ms_per_step'], d['steps']))
