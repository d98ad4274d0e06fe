"""Host cost of the engine's stream fork (vamd_stream_wait: an event record and a wait) and of launches that alternate
between two streams: python tests/diag_stream_fork_gpu.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vision_amd import _lib
from vision_amd._lib import lib, call

dev = torch.device("cuda:0")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()


def host_us(fn, n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


def launch(stream):
    # a small library launch (a 4 MB copy)
    call("vamd_add_cast", x.data_ptr(), _lib.F32, None, _lib.F32, y.data_ptr(), _lib.F32, x.numel(), stream)


print("launch on one stream            %6.1f us" % host_us(lambda: launch(main.cuda_stream)))
state = [0]
def alt():
    state[0] ^= 1
    launch(side.cuda_stream if state[0] else main.cuda_stream)
print("launches alternating streams    %6.1f us" % host_us(alt))
print("vamd_stream_wait alone          %6.1f us" % host_us(lambda: call("vamd_stream_wait", side.cuda_stream, main.cuda_stream)))
def fork_launch():
    launch(main.cuda_stream)
    call("vamd_stream_wait", side.cuda_stream, main.cuda_stream)
    launch(side.cuda_stream)
print("launch, fork, launch on side    %6.1f us (three calls)" % host_us(fork_launch))
def torch_fork():
    side.wait_stream(main)
print("torch wait_stream alone         %6.1f us" % host_us(torch_fork))
