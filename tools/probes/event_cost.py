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




# ---- does a second queue that merely runs kernels (no events at all) slow the first one down?  The side stream gets a backlog of
# medium kernels first (so the host is out of the picture), then the main-stream chain is timed while the backlog drains.
big = torch.zeros(1 << 22, device=dev)          # 16 MB: ~10 us per add_
for side_n, label in ((0, 'idle'), (1500, 'busy')):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        for _ in range(side_n):
            big.add_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record(sa)
        for i in range(N):
            a.add_(1.0)
        e1.record(sa)
    torch.cuda.synchronize()
    print('main-stream chain with the side queue %s: %.2f us per kernel' % (label, e0.elapsed_time(e1) * 1e3 / N))


# ---- the same hand-off with events that skip the system-scope fence (hipEventDisableSystemFence / hipEventReleaseToDevice): torch's
# events release to system scope when they are recorded; consumers on the same GPU only need device scope
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
for flags, label in ((0x2, 'DisableTiming (torch default)'), (0x2 | 0x20000000, 'DisableTiming | DisableSystemFence'),
                     (0x2 | 0x40000000, 'DisableTiming | ReleaseToDevice')):
    evs = []
    for _ in range(N):
        e = ctypes.c_void_p()
        rc = hip.hipEventCreateWithFlags(ctypes.byref(e), flags)
        assert rc == 0, rc
        evs.append(e)
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record(sa)
            for i in range(N):
                a.add_(1.0)
                assert hip.hipEventRecord(evs[i], sa.cuda_stream) == 0
                assert hip.hipStreamWaitEvent(sb.cuda_stream, evs[i], 0) == 0
                with torch.cuda.stream(sb):
                    b.add_(1.0)
            e1.record(sa)
        torch.cuda.synchronize()
    print('hand-off with %-40s %.2f us per main-stream kernel' % (label + ':', e0.elapsed_time(e1) * 1e3 / N))
