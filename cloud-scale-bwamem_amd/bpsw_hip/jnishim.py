"""Harness for the JNI shim (csrc/bpsw_jni.cpp) outside a JVM: the exported Java_* symbols are driven through the fake JNIEnv of
tests/fake_jvm/fake_jni.cpp, and the shim's own clock (bpsw_jni_last_times: marshalling in / the C ABI call / marshalling out of
the last call on the thread) tells what the marshalling costs at the reference's batch sizes.  Not part of the product path.

What the figures mean: for swExtendFPGAJNI the fake env's GetByteArrayRegion / SetShortArrayRegion are memcpys, as HotSpot's are
-- representative.  For mateSWJNI the fake env's objects are heap-allocated structs with slot-indexed fields (an index per access,
like an offset in a JVM, but a `new` with three vectors per AllocObject): its ns-per-region figures are an UPPER bound on a real
JVM for the field traffic and the allocation, and say nothing about GC or safepoints."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import LIB_PATH, ALNREG_DTYPE, default_opt, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_jvm")
FAKE_SO = os.path.join(FAKE_DIR, "libfakejvm.so")


def load_fake():
    src = os.path.join(FAKE_DIR, "fake_jni.cpp")
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(src):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "cloud-scale-bwamem_amd", "csrc"),
                        "-o", FAKE_SO, src, "-ldl"], check=True)
    lib = load_library()   # torch first (one HIP runtime per process), then the product library
    fake = C.CDLL(FAKE_SO)
    fake.fake_jvm_extend.restype = C.c_int
    fake.fake_jvm_matesw.restype = C.c_int
    fake.fake_jvm_matesw_flat.restype = C.c_int
    lib.bpsw_jni_last_times.argtypes = [C.POINTER(C.c_double)]
    lib.bpsw_jni_last_times.restype = None
    return fake, lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def last_times(lib):
    t = (C.c_double * 4)()
    lib.bpsw_jni_last_times(t)
    return list(t)


def extend(fake, wire, n, partition=-1):
    out = np.zeros(max(10 * n, 1), np.int16)
    err = C.create_string_buffer(512)
    rc = fake.fake_jvm_extend(LIB_PATH.encode(), partition, _vp(wire), int(wire.size), 10 * n, _vp(out), err, 512)
    return rc, out[: 10 * n], err.value.decode()


def matesw(fake, g, partition=-1, pac=None):
    opt = default_opt()
    ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5, opt.pen_clip3,
                     opt.w, opt.zdrop, opt.T, opt.flag, opt.min_seed_len, opt.max_ins, opt.max_matesw], np.int32)
    mat = np.array(list(opt.mat), np.int8)
    pes = np.array([[p[0], p[1], p[2], p[3], p[4]] for p in g.pes], np.float64)
    out_cnt = np.zeros(2 * g.group_size + 1, np.int32)
    cap = int(g.regs.shape[0] + g.ref_rb.shape[0] + 16)
    out = np.zeros(cap, ALNREG_DTYPE)
    total, frames = C.c_int64(0), C.c_long(0)
    err = C.create_string_buffer(512)
    rc = fake.fake_jvm_matesw(LIB_PATH.encode(), partition, _vp(ints), C.c_float(opt.mask_level_redun), _vp(mat),
                              C.c_int64(g.l_pac), _vp(pes), g.group_size, _vp(g.seq_len), _vp(g.seq_off), _vp(g.seq_pool),
                              _vp(g.reg_cnt), _vp(g.regs), _vp(g.ref_cnt), _vp(g.ref_rb), _vp(g.ref_re), _vp(g.ref_len),
                              _vp(g.ref_off), _vp(g.ref_pool), _vp(out_cnt), _vp(out), C.c_int64(cap), C.byref(total),
                              C.byref(frames), err, 512, _vp(pac) if pac is not None else None)
    return rc, out_cnt[: 2 * g.group_size], out[: total.value], frames.value, err.value.decode()


def matesw_flat(fake, g, partition=-1, pac=None):
    """the flat-array entry mateSWFlatJNI (round 4) on the same group; pac given: windows by coordinates, no window bytes"""
    opt = default_opt()
    ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5, opt.pen_clip3,
                     opt.w, opt.zdrop, opt.T, opt.flag, opt.min_seed_len, opt.max_ins, opt.max_matesw], np.int32)
    mat = np.array(list(opt.mat), np.int8)
    pes = np.array([[p[0], p[1], p[2], p[3], p[4]] for p in g.pes], np.float64)
    out_cnt = np.zeros(2 * g.group_size + 1, np.int32)
    cap = int(g.regs.shape[0] + g.ref_rb.shape[0] + 16)
    out = np.zeros(cap, ALNREG_DTYPE)
    total = C.c_int64(0)
    err = C.create_string_buffer(512)
    rc = fake.fake_jvm_matesw_flat(LIB_PATH.encode(), partition, _vp(ints), C.c_float(opt.mask_level_redun), _vp(mat),
                                   C.c_int64(g.l_pac), _vp(pes), g.group_size, _vp(g.seq_len), _vp(g.seq_off), _vp(g.seq_pool),
                                   _vp(g.reg_cnt), _vp(g.regs), _vp(g.ref_cnt), _vp(g.ref_rb), _vp(g.ref_re), _vp(g.ref_len),
                                   _vp(g.ref_off), _vp(g.ref_pool), _vp(out_cnt), _vp(out), C.c_int64(cap), C.byref(total),
                                   err, 512, _vp(pac) if pac is not None else None)
    return rc, out_cnt[: 2 * g.group_size], out[: total.value], err.value.decode()


def shim_rate(wire, n_tasks, group, reps=5, pac=None):
    """us per call (median of `reps` after one warm-up) of the two JNI symbols at the given batch, split at the C ABI"""
    fake, lib = load_fake()
    res = {}
    rows = []
    for k in range(reps + 1):
        rc, _, msg = extend(fake, wire, n_tasks)
        if rc != 0:
            raise RuntimeError("swExtendFPGAJNI through the fake JNIEnv: " + msg)
        if k:
            rows.append(last_times(lib))
    a = np.median(np.array(rows), axis=0)
    res["swExtendFPGAJNI"] = {"tasks_per_call": int(n_tasks), "wire_bytes": int(wire.size), "marshal_in_us": round(float(a[0]), 1),
                              "c_abi_call_us": round(float(a[1]), 1), "marshal_out_us": round(float(a[2]), 1),
                              "shim_share_of_call": round(float((a[0] + a[2]) / max(a[0] + a[1] + a[2], 1e-9)), 4)}
    if group is not None:
        rows = []
        for k in range(reps + 1):
            rc, _, regs, frames, msg = matesw(fake, group)
            if rc != 0:
                raise RuntimeError("mateSWJNI through the fake JNIEnv: " + msg)
            if k:
                rows.append(last_times(lib))
        a = np.median(np.array(rows), axis=0)
        n_in = int(group.regs.shape[0])
        res["mateSWJNI"] = {"pairs_per_call": int(group.group_size), "regions_in": n_in, "regions_out": int(a[3]),
                            "window_rows": int(group.ref_rb.shape[0] // 4), "marshal_in_us": round(float(a[0]), 1), "c_abi_call_us": round(float(a[1]), 1),
                            "marshal_out_us": round(float(a[2]), 1),
                            "marshal_in_ns_per_region_in": round(1e3 * float(a[0]) / max(n_in, 1), 1),
                            "marshal_out_ns_per_region_out": round(1e3 * float(a[2]) / max(a[3], 1), 1),
                            "shim_share_of_call": round(float((a[0] + a[2]) / max(a[0] + a[1] + a[2], 1e-9)), 4)}
    if group is not None:
        for key, use_pac in (("mateSWFlatJNI", None), ("mateSWFlatJNI_coordinates", pac)):
            if key.endswith("coordinates") and pac is None:
                continue
            rows = []
            for k in range(reps + 1):
                rc, _, regs, msg = matesw_flat(fake, group, pac=use_pac)
                if rc != 0:
                    raise RuntimeError(key + " through the fake JNIEnv: " + msg)
                if k:
                    rows.append(last_times(lib))
            a = np.median(np.array(rows), axis=0)
            res[key] = {"pairs_per_call": int(group.group_size), "regions_in": int(group.regs.shape[0]), "regions_out": int(a[3]),
                        "marshal_in_us": round(float(a[0]), 1), "c_abi_call_us": round(float(a[1]), 1), "marshal_out_us": round(float(a[2]), 1),
                        "shim_share_of_call": round(float((a[0] + a[2]) / max(a[0] + a[1] + a[2], 1e-9)), 4)}
    res["note"] = ("fake JNIEnv (tests/fake_jvm): array regions are memcpys as in HotSpot; objects are heap-allocated C++ structs with "
                   "slot-indexed fields and every Get<Type>Field goes through the function table, so the per-region figures of mateSWJNI are "
                   "an upper bound on a JVM's field traffic and allocation, and exclude GC / safepoints; one calling thread")
    return res
