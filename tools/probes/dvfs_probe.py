import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from sniper_amd import hip
d = torch.device('cuda', 0)
B, H, W, C, O, K = 20, 32, 32, 256, 256, 3
x = (torch.randn(B, H, W, C, device=d) * 0.5).half(); w = (torch.randn(O, K * K, C, device=d) * 0.05).half()
y = torch.empty((B, H, W, O), dtype=torch.float16, device=d)
run = lambda: hip.call('sn_conv_fwd', x, w, None, None, y, B, H, W, C, C, O, O, O, K, K, 1, 1, 1, 0, 0, hip.stream())
for iters in (10, 100, 1000, 5000, 20000):
    run(); torch.cuda.synchronize(); time.sleep(0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    print('iters %6d  %.2f us/launch' % (iters, e0.elapsed_time(e1) / iters * 1e3), flush=True)
os.system('rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk" | head -4')
