"""Timing survey of the paths bench.py has no line for (one MI355X, 4096 streams): NS with bands
(32 / 48 kHz), the int16 NS step, the unfused Analyze / Process pair.  Wall clock around K queued
steps (device buffers), after a warm-up."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosignalprocess_amd.ns import NsBatch, _check
from audiosignalprocess_amd._abi import MEM_DEVICE
from audiosignalprocess_amd.synth import ns_frames

S, F = 4096, 60
x = torch.from_numpy(ns_frames(S, F, frame0=0)).cuda()


def timed(fn, sync, steps=200, warm=50):
    for k in range(warm): fn(k)
    sync()
    t0 = time.perf_counter()
    for k in range(steps): fn(k)
    sync()
    return (time.perf_counter() - t0) / steps * 1e6


for fs, nh in ((32000, 1), (48000, 2)):
    g = NsBatch(S, fs=fs, policy=1)
    hi = (0.1 * x[:, None, :, :]).repeat(1, nh, 1, 1).contiguous()
    ol, oh = torch.empty_like(x), torch.empty_like(hi)
    lib = g.lib
    def step(k, g=g, hi=hi, ol=ol, oh=oh):
        f = k % F
        _check(lib.AspNsBatch_AnalyzeProcessBands(g.h, C.c_void_p(x[f].data_ptr()), C.c_void_p(hi[f].data_ptr()),
                                                  C.c_void_p(ol[f].data_ptr()), C.c_void_p(oh[f].data_ptr()), 1, MEM_DEVICE), "bands")
    us = timed(step, g.synchronize)
    print("NS %d Hz (low band + %d high band(s)): %.1f us per frame step of %d streams" % (fs, nh, us, S))
    g.close()

g = NsBatch(S, policy=1)
y = torch.empty_like(x)
def fused(k):
    f = k % F
    g.analyze_process_device(x[f].data_ptr(), y[f].data_ptr(), 1)
print("NS 16 kHz fused, one call per frame: %.1f us" % timed(fused, g.synchronize))
x16 = x.to(torch.int16); y16 = torch.empty_like(x16)
def s16(k):
    f = k % F
    _check(g.lib.AspNsBatch_AnalyzeProcessS16(g.h, C.c_void_p(x16[f].data_ptr()), C.c_void_p(y16[f].data_ptr()), 1, MEM_DEVICE), "s16")
print("NS 16 kHz fused, int16 PCM: %.1f us" % timed(s16, g.synchronize))
def unfused(k):
    f = k % F
    _check(g.lib.AspNsBatch_Analyze(g.h, C.c_void_p(x[f].data_ptr()), MEM_DEVICE), "analyze")
    _check(g.lib.AspNsBatch_Process(g.h, C.c_void_p(x[f].data_ptr()), C.c_void_p(y[f].data_ptr()), MEM_DEVICE), "process")
print("NS 16 kHz Analyze + Process as two calls: %.1f us" % timed(unfused, g.synchronize))
g.close()
