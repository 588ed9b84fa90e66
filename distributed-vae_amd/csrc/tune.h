// Private: indices of mmvae_exec.tune (include/mmvae.h) -- switches between LIVE code paths (other shapes or engines take them
// anyway) for A/B timing, diagnostic stamps and test hooks.  0 = production behaviour for every one of them.  MMVAE_TUNE_ENGINE
// (17) is public and defined in mmvae.h.  The Python binding translates environment variables into these (_native.TUNE_ENV); the
// library itself reads none.  Switches whose experiment is settled are gone with their code (round 4: where dW11 forks, extra
// LDS for dW11, small-layer products on the side stream, the coupling behind the dW11 fork, the join behind the last reduction,
// recorded fork events, the fc11 tensors' reduction on the main stream, smaller forward chain blocks, one wave per cell in the
// latent kernels, the one-launch encoder chains: numbers in DESIGN.md appendix, code of the last in tools/experiments/).
#pragma once
enum {
    MMVAE_TUNE_EVAL_CHAIN_OFF = 0, // eval mode: fc2..fc5 as four launches instead of one
    MMVAE_TUNE_AUG_TILE = 3,       // augmenter GEMMs: fp32 matrix instruction: tile 11 12 21 22 (1 = 64, 2 = 128); planes x planes engine:
                                   // 1 / 2 / 3 = 256 x 256 / 256 x 128 / 128 x 128 (+ 10 KS: K split); 90 = the tile engine of gemm_bf16.hip
    MMVAE_TUNE_ABLATE_C = 4,       // chain kernels: timing ablations / cycle stamps (bit 3: stamps; results wrong with bits 0..2)
    MMVAE_TUNE_ABLATE = 5,         // fc1 forward ablations (fp32 matrix-instruction kernels)
    MMVAE_TUNE_FC11_ZG_OFF = 8,    // fc11 forward, loss and d(d10) as separate launches instead of the fused kernel
    MMVAE_TUNE_COUPLE_SIDE = 13,   // fused step, where the coupling terms run: 0 = as a role of the decoder chain's launch from four arms up
                                   // and on the side stream below (chain.hip dec_couple_ok), 1 = side stream always, 3 = role always,
                                   // 2 = the role's launch with its workgroups exiting at once (timing experiment, results wrong)
    MMVAE_TUNE_ABLATE_L = 14,      // latent kernels: ablations / stamps
    MMVAE_TUNE_ABLATE_B = 16,      // bf16 GEMM engine: 1 no MFMAs, 2 no global loads, 4 no LDS stores (results wrong)
    // 17 MMVAE_TUNE_ENGINE: public (mmvae.h)
    MMVAE_TUNE_BF16_NARROW_FP32 = 18, // bf16 configuration on bf16 storage: the narrow operands of fc1 / dW1 (W1, dZ1) read as fp32 and rounded
                                   // by every block tile instead of as bf16 from slice 0 of their planes
    MMVAE_TUNE_BN_PARTIALS = 19,   // BatchNorm batch sums through per-workgroup partial arrays instead of the accumulators
    MMVAE_TUNE_PRESPLIT_ALL = 20,  // fp32x3 engine: all slice planes through k_presplit launches
    MMVAE_TUNE_CHAIN_FP32 = 21,    // fp32x3 engine: the chain kernels' own GEMMs stay on the fp32 matrix instruction
};
