"""What hipcc emits for the headline kernel, looked at: no GPU needed (hipcc cross-compiles gfx950), seconds."""
import os
import subprocess

def test_trunk_kernel_has_no_packed_fp32_instructions(tmp_path):
    """A packed-fp32 instruction gets no issue slot beside the other wave's MFMAs (profiles/r05_coissue/): trunkw_kernel is built
    with __attribute__((target("no-packed-fp32-ops"))) (UVA_NO_PK_F32, csrc/uva_devutil.hip.h).  Should a compiler stop honouring
    that, the library would still be correct and 2 % slower without anybody noticing: the assembly hipcc emits HERE is looked at."""
    from upscale_video_amd import build
    asm = str(tmp_path / "uva_wino.s")
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "uva_wino.hip"), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(asm).read()
    body = text[text.index("trunkw_kernel"):]
    mfma = body.count("v_mfma_f32_16x16x32_f16")
    packed = sum(body.count(op) for op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"))
    assert mfma >= 3 * 192, mfma            # three instantiations, two k-loops of 96 each
    assert packed == 0, packed


def test_winograd_conv5_kernel_keeps_what_round_6_fixed(tmp_path):
    """g_conv3_sww (csrc/uva_sww.hip.h): 288 MFMAs per block and instantiation (two thirds of the direct kernel's 432), no
    scratch, the weights read by the MFMAs straight from AccVGPRs (left alone hipcc parks the overflow of the 288 weight
    registers there behind a v_accvgpr_read in front of every use: 267 extra VALU instructions per block) and loaded in one
    pipelined burst (pinned one by one behind its own load every weight waited for its own memory round trip).  None of this
    changes a result, so only the assembly can tell whether a compiler update kept it."""
    import re
    from upscale_video_amd import build
    asm = str(tmp_path / "uva_sww.s")
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "uva_sww.hip"), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(asm).read()
    kernels = re.findall(r"^(_ZN3uva11g_conv3_swwILi\dELi\dELb[01]EEEvNS_7GSwArgsE):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 5, [k for k, _ in kernels]       # (no sum; one / two sums with the first operand from HBM or from the LDS ring)
    for name, body in kernels:
        mfma = re.findall(r"v_mfma_f32_16x16x32_f16 [^\n]*", body)
        assert len(mfma) == 288, (name, len(mfma))
        from_acc = sum(1 for m in mfma if re.match(r"v_mfma_f32_16x16x32_f16 [av]\[\d+:\d+\], a\[", m))
        assert from_acc >= 170, (name, from_acc)                      # (the 44 pinned k-steps' 176 MFMAs; the rest have theirs in ArchVGPRs)
        assert body.count("v_accvgpr_read") <= 140, (name, body.count("v_accvgpr_read"))            # (84-103 today; 267 unpinned)
        assert "scratch_" not in body, name
        assert body.count("s_waitcnt vmcnt(0)") <= 8, (name, body.count("s_waitcnt vmcnt(0)"))      # (the weights arrive in one burst)
    assert not re.search(r"ScratchSize: [1-9]", text)
