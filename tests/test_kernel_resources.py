"""CPU: register budgets of the hot kernels, from the compiler's own resource remarks (kept by the build beside each object).
Found the hard way in round 2: an epilogue loop nest too big to unroll fully makes hipcc index the MFMA accumulators dynamically,
and the whole accumulator tile moves to scratch memory (528 bytes per lane) -- correct results, a fraction of the speed, nothing in
any parity test notices.  This test does."""
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, 'sniper_amd', 'lib', 'obj')


def _resources(unit):
    """{mangled kernel name: {resource: value}} from the remarks sniper_amd/build.py keeps beside the object file"""
    path = os.path.join(OBJ, unit + '.remarks')
    if not os.path.isfile(path):
        pytest.skip('%s not built here (python -m sniper_amd.build writes it)' % path)
    out, name = {}, None
    with open(path) as fh:
        for ln in fh:
            m = re.search(r'Function Name: (\S+)', ln)
            if m:
                name = m.group(1)
                out[name] = {}
                continue
            m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', ln)
            if m and name:
                out[name][m.group(1).strip()] = int(m.group(2))
    return out


def test_hot_kernels_keep_their_accumulators_in_registers():
    conv, roi = _resources('conv_dma'), _resources('roi_deform')
    dma = {k: v for k, v in conv.items() if 'conv_dma_kernel' in k}
    assert len(dma) == 2 * 9, sorted(dma)                        # forward + data gradient of every configuration (conv_dma.hip kCfg)
    for name, res in dma.items():
        # (round 6: also the 256 x 256 tile -- 8 waves at the 256-VGPR cap -- keeps everything in registers: 28 B / lane until then)
        assert res['ScratchSize'] == 0 and res['VGPRs Spill'] == 0, (name, res)
        assert res['VGPRs'] <= 256, (name, res)
    # the persistent twins (PERSIST = the last template argument): two workgroups per CU like the configurations they stand in for --
    # 8-wave workgroups need <= 128 registers (4 waves per SIMD); a spill there would put tracked scratch loads, hence queue drains,
    # into the tile loop (conv_dma.hip)
    persist = {k: v for k, v in dma.items() if k.endswith('ELb0ELb1EEv10ConvParamsii')}
    assert len(persist) == 4, sorted(persist)
    for name, res in persist.items():
        waves8 = 'ELi2ELi4ELi2E' in name
        assert res['VGPRs'] <= (128 if waves8 else 256) and res['Occupancy'] >= (4 if waves8 else 2), (name, res)
    ps = {k: v for k, v in _resources('conv_wgrad_ps').items() if 'wgrad_ps_kernel' in k}
    assert ps, 'wgrad_ps_kernel not built'
    for name, res in ps.items():        # 64 accumulator + 64 fragment registers per consumer wave; the 12-wave form (256 x 128 tiles:
        # 8 consumers + 4 producers) has three waves per SIMD, i.e. at most 168 registers each
        wide = 'ELi3ELi2E' in name
        assert res['ScratchSize'] == 0 and res['VGPRs Spill'] == 0 and res['VGPRs'] <= (168 if wide else 256), (name, res)
    for frag in ('dpsroi_bwd_data_mfma_kernel', 'deform_col2im_data_mfma_kernel', 'dpsroi_fwd_roi_kernel'):
        hits = {k: v for k, v in roi.items() if frag in k}
        assert hits, frag
        for name, res in hits.items():
            assert res['ScratchSize'] == 0 and res['VGPRs Spill'] == 0, (name, res)
    # the offset gradient of the deformable PS-RoI pooling is DELIBERATELY capped at 64 VGPRs (8 waves per SIMD; the spilled values
    # belong to the once-per-RoI geometry phase): 253 -> 211 us, profiles/r04_kab_roi_occupancy.txt
    for name, res in roi.items():
        if 'dpsroi_bwd_trans_roi_kernel' in name:
            assert res['VGPRs'] <= 64 and res['Occupancy'] == 8 and res['ScratchSize'] <= 256, (name, res)
    # the matrix-core gathers are launched five workgroups deep per CU (1280 tile workgroups on 256 CUs): <= 96 VGPRs
    for name, res in roi.items():
        if 'dpsroi_bwd_data_mfma_kernel' in name:
            assert res['VGPRs'] <= 96 and res['Occupancy'] >= 5, res
