"""What does a cross-stream event (record on the main stream + wait on the side stream) cost the main stream?  Chains of small kernels
on stream A with / without an event hand-off to stream B after each one."""
import sys, time, torch
dev = torch.device('cuda:0')
a = torch.zeros(1 << 16, device=dev)
b = torch.zeros(1 << 16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 400


def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record(sa)
        for i in range(N):
            a.add_(1.0)
            if mode in ('event', 'event_nowork'):
                ev = torch.cuda.Event()
                ev.record(sa)
                sb.wait_event(ev)
                if mode == 'event':
                    with torch.cuda.stream(sb):
                        b.add_(1.0)
            elif mode == 'record_only':
                ev = torch.cuda.Event()
                ev.record(sa)
        e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


for mode in ('plain', 'record_only', 'event_nowork', 'event', 'plain'):
    run(mode)
    print('%-14s %.2f us per main-stream kernel' % (mode, run(mode)))

