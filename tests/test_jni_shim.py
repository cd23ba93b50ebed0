"""Drives the exported Java_* symbols of libbPSW_hip.so through a fake JNIEnv (tests/fake_jvm/fake_jni.cpp): the
argument graphs are built the way memSamPeGroupJNIPrepare / runOnFPGAJNI build them, the result is flattened the way
mateSWArrayToAlnRegPairArray reads it.  No JVM exists in the image, so this is the closest available check of the
marshalling code; the JNI function-table slot numbers are taken from the JNI specification (csrc/jni_min.h)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
from conftest import region_fields_equal

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE_SO = os.path.join(HERE, "fake_jvm", "libfakejvm.so")


from bpsw_hip import jnishim


@pytest.fixture(scope="module")
def fake():
    return jnishim.load_fake()[0]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _extend(fake, wire, n, partition=-1):
    return jnishim.extend(fake, wire, n, partition)


def _matesw(fake, g, partition=-1, pac=None):
    return jnishim.matesw(fake, g, partition, pac)


def test_without_a_device_the_shim_raises_a_java_exception(fake):
    if bpsw_hip.load_library().bpsw_device_count() > 0:
        pytest.skip("a GPU is visible")
    soa = synth.ext_tasks(16, seed=3)
    rc, _, msg = _extend(fake, bpsw_hip.wire_pack(soa), soa.n)
    assert rc == 1 and msg.startswith("java/lang/RuntimeException: bPSW: no usable HIP device")
    rc, _, _, frames, msg = _matesw(fake, synth.rescue_group(4, seed=3, p_resc=0.5))
    assert rc == 1 and "RuntimeException" in msg and frames == 0       # every PushLocalFrame was popped


@pytest.mark.gpu
def test_shim_clock_splits_a_call_at_the_c_abi(fake, ctx):
    """bpsw_jni_last_times (the shim micro-benchmark's clock): the three parts are measured and the marshalling of a wire batch --
    one GetByteArrayRegion into pinned staging, one SetShortArrayRegion out of the pinned result block -- stays a small part"""
    soa = synth.ext_tasks(8000, seed=78)
    grp = synth.rescue_group(200, seed=79, p_resc=0.3)
    r = jnishim.shim_rate(bpsw_hip.wire_pack(soa), soa.n, grp, reps=3)
    e, m = r["swExtendFPGAJNI"], r["mateSWJNI"]
    assert e["c_abi_call_us"] > 0 and e["marshal_in_us"] > 0 and e["marshal_out_us"] >= 0
    assert e["shim_share_of_call"] < 0.5
    assert m["c_abi_call_us"] > 0 and m["regions_out"] >= m["regions_in"] > 0


@pytest.mark.gpu
def test_extend_through_jni_matches_c_abi(fake, ctx):
    soa = synth.ext_tasks(2000, seed=77)
    wire = bpsw_hip.wire_pack(soa)
    want = ctx.extend_batch(wire)
    for partition in (-1, 5):          # no TaskContext / TaskContext.get().partitionId() == 5
        rc, got, msg = _extend(fake, wire, soa.n, partition)
        assert rc == 0, msg
        assert np.array_equal(got, want)
    rc, _, msg = _extend(fake, wire[: 32 + 32 * soa.n - 8], soa.n)
    assert rc == 1 and "RuntimeException" in msg                        # malformed batch -> exception, not a crash


# (read_len 250: BASELINE.json configs[4] -- 8 % / 2 % error, the packed kernel's five-columns-per-lane build, the ring's second class)
RL_CASES = [dict(), dict(read_len=250, sub_rate=0.08, indel_rate=0.02)]


@pytest.mark.gpu
@pytest.mark.parametrize("rl", RL_CASES, ids=["150bp", "250bp"])
@pytest.mark.parametrize("allo", [False, True])
def test_matesw_through_jni_matches_c_abi(fake, ctx, allo, rl):
    g = synth.rescue_group(120, seed=404, p_resc=0.4, all_orientations=allo, **rl)
    want_cnt, want = ctx.matesw_group(bpsw_hip.default_opt(), g)
    rc, cnt, regs, frames, msg = _matesw(fake, g, partition=3)
    assert rc == 0, msg
    assert frames == 0
    assert np.array_equal(cnt, want_cnt)
    region_fields_equal(regs, want)


@pytest.mark.gpu
def test_matesw_lazy_path_steps_aside_for_arrays_in_another_order(fake, ctx, monkeypatch):
    """Round 5: mateSWJNI unmarshals only the pairs that may need a job and hands the caller's own objects back for the others -- relying on
    the order memSamPeGroupJNIPrepare builds its arrays in (MemSamPe.scala:1931-1990).  The contract itself keys every object by its own
    indices: with RefSWType[] and SeqSWType[] reversed the lazy path must notice and the eager walk (any order) must give the same answer."""
    lib = bpsw_hip.load_library()
    lib.bpsw_jni_last_mate_path.restype = C.c_int
    g = synth.rescue_group(140, seed=407, p_resc=0.4, p_multi_anchor=0.3)
    want_cnt, want = ctx.matesw_group(bpsw_hip.default_opt(), g)
    rc, cnt, regs, frames, msg = _matesw(fake, g, partition=2)
    assert rc == 0 and frames == 0, msg
    assert lib.bpsw_jni_last_mate_path() == 1                     # the lazy path
    assert np.array_equal(cnt, want_cnt); region_fields_equal(regs, want); assert np.array_equal(regs["hash"], want["hash"])
    monkeypatch.setenv("FAKE_JVM_SHUFFLE", "1")
    rc, cnt, regs, frames, msg = _matesw(fake, g, partition=2)
    assert rc == 0 and frames == 0, msg
    assert lib.bpsw_jni_last_mate_path() == 2                     # stepped aside
    assert np.array_equal(cnt, want_cnt); region_fields_equal(regs, want); assert np.array_equal(regs["hash"], want["hash"])
    # a group in which nothing needs rescue: no SW job, no new object
    g0 = synth.rescue_group(60, seed=408, p_resc=0.0)
    monkeypatch.delenv("FAKE_JVM_SHUFFLE")
    rc, cnt, regs, frames, msg = _matesw(fake, g0, partition=2)
    assert rc == 0 and frames == 0, msg
    assert lib.bpsw_jni_last_mate_path() == 1
    assert np.array_equal(cnt, g0.reg_cnt) and np.array_equal(regs["rb"], g0.regs["rb"]) and np.array_equal(regs["hash"], g0.regs["hash"])


@pytest.mark.gpu
def test_matesw_through_jni_with_reference_on_device(fake, ctx):
    """SURVEY.md 8f.2 through the JNI surface: loadPacJNI, then RefSWType objects that carry (rBeg, rEnd) only"""
    l_pac = 300_007
    pac, bases = synth.random_pac(l_pac, seed=51)
    g = synth.rescue_group(150, seed=52, l_pac=l_pac, p_resc=0.4, ref_bases=bases)
    want_cnt, want = ctx.matesw_group(bpsw_hip.default_opt(), g)          # windows shipped as bytes
    rc, cnt, regs, frames, msg = _matesw(fake, g, partition=1, pac=pac)  # windows named by coordinates
    assert rc == 0, msg
    assert frames == 0
    assert np.array_equal(cnt, want_cnt)
    region_fields_equal(regs, want)
    assert regs.shape[0] > g.regs.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("rl", RL_CASES, ids=["150bp", "250bp"])
@pytest.mark.parametrize("allo", [False, True])
def test_matesw_flat_entry_equals_the_object_array_entry(fake, ctx, allo, rl):
    """mateSWFlatJNI (round 4: primitive arrays in, one long[] out) against mateSWJNI on the same group, and against the C ABI"""
    g = synth.rescue_group(160, seed=811, p_resc=0.4, all_orientations=allo, **rl)
    want_cnt, want = ctx.matesw_group(bpsw_hip.default_opt(), g)
    rc, cnt_o, regs_o, frames, msg = _matesw(fake, g, partition=2)
    assert rc == 0, msg
    rc, cnt_f, regs_f, msg = jnishim.matesw_flat(fake, g, partition=2)
    assert rc == 0, msg
    assert np.array_equal(cnt_f, cnt_o) and np.array_equal(cnt_f, want_cnt)
    region_fields_equal(regs_f, regs_o)
    region_fields_equal(regs_f, want)
    assert np.array_equal(regs_f["hash"], regs_o["hash"])


@pytest.mark.gpu
def test_matesw_flat_entry_with_reference_on_device(fake, ctx):
    """the flat entry with refLen = refBytes = null: windows by (rBeg, rEnd) from the reference loaded with loadPacJNI"""
    l_pac = 300_007
    pac, bases = synth.random_pac(l_pac, seed=53)
    g = synth.rescue_group(150, seed=54, l_pac=l_pac, p_resc=0.4, ref_bases=bases)
    want_cnt, want = ctx.matesw_group(bpsw_hip.default_opt(), g)          # windows shipped as bytes
    rc, cnt, regs, msg = jnishim.matesw_flat(fake, g, partition=1, pac=pac)
    assert rc == 0, msg
    assert np.array_equal(cnt, want_cnt)
    region_fields_equal(regs, want)
    assert regs.shape[0] > g.regs.shape[0]


@pytest.mark.gpu
def test_chain2aln_through_jni_matches_c_abi(fake, ctx):
    """SURVEY.md 8f.3 through the JNI surface: loadPacJNI + chainToAlnJNI (primitive arrays only)"""
    l_pac = 300_007
    pac, bases = synth.random_pac(l_pac, seed=81)
    b = synth.read_chains(600, bases, l_pac, read_len=150, sub_rate=0.02, indel_rate=0.004, seed=82)
    ctx.ref_load(pac, l_pac)
    opt = bpsw_hip.default_opt()
    want_cnt, want = ctx.chain2aln_batch(opt, b)
    reads = np.concatenate([b.read_pool[o:o + n] for o, n in zip(b.read_off, b.read_len)])   # back to back, as the Scala side sends them
    ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_clip5, opt.pen_clip3, opt.w, opt.zdrop], np.int32)
    mat = np.array(list(opt.mat), np.int8)
    out = np.zeros(b.n_reads + 8 * len(b.seed_len) + 8, np.int64)
    out_n = C.c_int64(0)
    err = C.create_string_buffer(512)
    fake.fake_jvm_chain2aln.restype = C.c_int
    rc = fake.fake_jvm_chain2aln(bpsw_hip.LIB_PATH.encode(), 2, _vp(pac), C.c_int64(l_pac), _vp(ints), _vp(mat), b.n_reads, _vp(b.read_len),
                                 _vp(reads), C.c_int64(reads.size), _vp(b.chain_cnt), C.c_int64(len(b.seed_cnt)), _vp(b.seed_cnt),
                                 C.c_int64(len(b.seed_len)), _vp(b.seed_rbeg), _vp(b.seed_qbeg), _vp(b.seed_len), _vp(out),
                                 C.c_int64(out.size), C.byref(out_n), err, 512)
    assert rc == 0, err.value.decode()
    n = b.n_reads
    assert out_n.value == n + 8 * len(want)
    assert np.array_equal(out[:n], want_cnt)
    flat = out[n:out_n.value].reshape(-1, 8)
    for k, f in enumerate(("rb", "re", "qb", "qe", "score", "truesc", "w", "seedcov")):
        assert np.array_equal(flat[:, k], want[f].astype(np.int64)), f


@pytest.mark.gpu
def test_sam_pe_tail_through_jni_matches_c_abi(fake, ctx, orc):
    """SURVEY.md 8f.1 / 8f.4 through the JNI surface: loadPacJNI + loadBnsJNI + samPeTailJNI (primitive arrays in, SAM text out)"""
    from tail_util import synthetic_group
    pac, g = synthetic_group(orc, 250, 7070, sub_rate=0.03, indel_rate=0.008, p_span=0.05)
    names = [bytes(g.ann_name_pool[int(g.ann_name_off[i]):int(g.ann_name_off[i + 1])]).decode() for i in range(g.ann_off.shape[0])]
    ctx.ref_load(pac, g.l_pac)
    ctx.bns_load(g.ann_off, g.ann_len, names)
    opt, topt = bpsw_hip.default_opt(), bpsw_hip.default_tail_opt()
    want, _ = ctx.sam_pe_batch(opt, topt, g)
    n2 = 2 * g.group_size
    reads = np.concatenate([g.read_pool[o:o + n] for o, n in zip(g.read_off, g.read_len)])      # back to back, as the Scala side sends them
    quals = np.concatenate([g.qual_pool[o:o + n] for o, n in zip(g.read_off, g.read_len)])
    name_len = np.diff(g.name_off).astype(np.int32)
    rnames = np.ascontiguousarray(g.name_pool[:int(g.name_off[-1])])
    ann_names = np.frombuffer(b"".join(n.encode() + b"\0" for n in names), np.uint8).copy()
    ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.w, opt.T, opt.flag, opt.min_seed_len,
                     topt.mapq_coef_fac], np.int32)
    reals = np.array([topt.mask_level, topt.mapq_coef_len] + [float(v) for p in g.pes for v in p], np.float64)
    mat = np.array(list(opt.mat), np.int8)
    reg_longs = np.ascontiguousarray(np.stack([g.regs["rb"], g.regs["re"]], axis=1).reshape(-1), np.int64)
    reg_ints = np.ascontiguousarray(np.stack([g.regs[f] for f in ("qb", "qe", "score", "truesc", "sub", "csub", "sub_n", "w", "seedcov", "secondary")],
                                             axis=1).reshape(-1), np.int32)
    out = np.zeros(sum(len(t) for t in want) + 64, np.uint8)
    out_off = np.zeros(n2 + 1, np.int64)
    nb = C.c_int64(0)
    err = C.create_string_buffer(512)
    fake.fake_jvm_sam_pe_tail.restype = C.c_int
    rc = fake.fake_jvm_sam_pe_tail(bpsw_hip.LIB_PATH.encode(), 3, _vp(pac), C.c_int64(g.l_pac), C.c_int(len(names)), _vp(g.ann_off), _vp(g.ann_len),
                                   _vp(ann_names), C.c_int64(ann_names.size), _vp(ints), _vp(reals), _vp(mat), C.c_int64(g.id0), C.c_int(n2),
                                   _vp(np.ascontiguousarray(g.read_len)), _vp(reads), _vp(quals), C.c_int64(reads.size), _vp(name_len), _vp(rnames),
                                   C.c_int64(rnames.size), _vp(np.ascontiguousarray(g.reg_cnt)), _vp(reg_longs), _vp(reg_ints),
                                   C.c_int64(g.regs.shape[0]), _vp(out), C.c_int64(out.size), C.byref(nb), _vp(out_off), err, 512)
    assert rc == 0, err.value.decode()
    text = out[:nb.value].tobytes()
    got = [text[int(out_off[i]):int(out_off[i + 1])] for i in range(n2)]
    assert got == want
    # the same call in two halves (round 5): samPeTailSubmitJNI x 6 from this one thread, samPeTailCollectJNI in reverse order -- the
    # library's tail workers do the plan, the kernel and the text; every copy gives the text above, a handle is good for one collect
    out[:] = 0
    out_off[:] = 0
    fake.fake_jvm_sam_pe_tail_async.restype = C.c_int
    rc = fake.fake_jvm_sam_pe_tail_async(bpsw_hip.LIB_PATH.encode(), 3, _vp(pac), C.c_int64(g.l_pac), C.c_int(len(names)), _vp(g.ann_off), _vp(g.ann_len),
                                         _vp(ann_names), C.c_int64(ann_names.size), _vp(ints), _vp(reals), _vp(mat), C.c_int64(g.id0), C.c_int(n2),
                                         _vp(np.ascontiguousarray(g.read_len)), _vp(reads), _vp(quals), C.c_int64(reads.size), _vp(name_len), _vp(rnames),
                                         C.c_int64(rnames.size), _vp(np.ascontiguousarray(g.reg_cnt)), _vp(reg_longs), _vp(reg_ints),
                                         C.c_int64(g.regs.shape[0]), _vp(out), C.c_int64(out.size), C.byref(nb), _vp(out_off), err, 512, C.c_int(6))
    assert rc == 0, err.value.decode()
    text = out[:nb.value].tobytes()
    assert [text[int(out_off[i]):int(out_off[i + 1])] for i in range(n2)] == want


@pytest.mark.gpu
def test_class_and_field_ids_are_resolved_once(fake, ctx):
    """native/jni_mate_sw.c:102-143 looks ~55 fields and 6 classes up by name on every call; the shim does it on the first call
    only (classes held as global references) -- the second mateSWJNI makes no FindClass / GetFieldID / GetMethodID at all"""
    fake.fake_jvm_lookups.restype = C.c_long
    fake.fake_jvm_global_refs.restype = C.c_long
    g = synth.rescue_group(40, seed=405, p_resc=0.4)
    rc, cnt0, regs0, _, msg = _matesw(fake, g, partition=2)
    assert rc == 0, msg
    before = fake.fake_jvm_lookups()
    rc, cnt1, regs1, frames, msg = _matesw(fake, g, partition=2)
    assert rc == 0, msg
    assert fake.fake_jvm_lookups() == before
    assert fake.fake_jvm_global_refs() >= 6
    assert frames == 0 and np.array_equal(cnt0, cnt1)
    region_fields_equal(regs0, regs1)


@pytest.mark.gpu
def test_eight_task_threads_partition_to_device_slots(fake, ctx):
    """Multi-GPU readiness on a one-GPU box: BPSW_DEVICES lists four entries (all the same physical device here;
    on an 8-GPU node they would be 0..7), eight task threads report partitions 0..7 through the fake TaskContext.  Every thread
    must land on entry `partition mod 4` of the list, own a context no other thread shares (one stream and one set of arenas per
    task thread), and compute the right answer while the others run."""
    lib = bpsw_hip.load_library()
    lib.bpsw_device_slots.restype = C.c_int
    lib.bpsw_device_for_partition.restype = C.c_int
    fake.fake_jvm_extend_threads.restype = C.c_int
    soa = synth.ext_tasks(3000, seed=78)
    wire = bpsw_hip.wire_pack(soa)
    want = ctx.extend_batch(wire)
    saved = os.environ.get("BPSW_DEVICES")
    os.environ["BPSW_DEVICES"] = "0,0,0,0"
    try:
        assert lib.bpsw_device_slots() == 4
        assert [lib.bpsw_device_for_partition(p) for p in range(8)] == [0] * 8 and lib.bpsw_device_for_partition(-1) == -1
        n = 8
        parts = np.arange(n, dtype=np.int32)
        outs = np.zeros((n, 10 * soa.n), np.int16)
        info = np.zeros((n, 4), np.int64)
        err = C.create_string_buffer(512)
        rc = fake.fake_jvm_extend_threads(bpsw_hip.LIB_PATH.encode(), n, _vp(parts), _vp(wire), int(wire.size), 10 * soa.n, 3,
                                          _vp(outs), _vp(info), err, 512)
        assert rc == 0, err.value.decode()
    finally:
        if saved is None:
            os.environ.pop("BPSW_DEVICES", None)
        else:
            os.environ["BPSW_DEVICES"] = saved
    for t in range(n):
        assert np.array_equal(outs[t], want)
        assert info[t, 0] == t and info[t, 1] == t % 4 and info[t, 2] == 0      # partition seen, BPSW_DEVICES entry, device
    assert len(set(int(h) for h in info[:, 3])) == n and 0 not in info[:, 3]     # eight distinct live contexts


def test_partition_slots_without_a_device():
    lib = bpsw_hip.load_library()
    if lib.bpsw_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert lib.bpsw_device_slots() == 0 and lib.bpsw_device_for_partition(3) == -1
