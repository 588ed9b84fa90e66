// Private: indices of mmvae_exec.tune (include/mmvae.h) -- experiment switches for A/B timing, ablations and test hooks.
// 0 = production behaviour for every one of them.  MMVAE_TUNE_ENGINE (17) is public and
// defined in mmvae.h.  The Python binding translates environment variables into these (_native.TUNE_ENV); the library
// itself reads none.
#pragma once
enum {
    MMVAE_TUNE_EVAL_CHAIN_OFF = 0, // eval mode: fc2..fc5 as four launches instead of one
    MMVAE_TUNE_DW11_AT = 1,        // where dW11 forks: 0 start of backward, 1 after decoder chain, 2 after latent, 3 not forked
    MMVAE_TUNE_SIDE_SMALL = 2,     // small-layer dW GEMMs on the side stream
    MMVAE_TUNE_AUG_TILE = 3,       // augmenter GEMM tile 11 12 21 22 (1 = 64, 2 = 128)
    MMVAE_TUNE_ABLATE_C = 4,       // chain kernels: timing ablations / cycle stamps (bit 3: stamps; results wrong with bits 0..2)
    MMVAE_TUNE_ABLATE = 5,         // fc1 forward ablations (fp32 matrix-instruction kernels)
    // 6: retired (was MMVAE_TUNE_PADLDS, extra dynamic LDS of the fp32 fc1 kernels)
    MMVAE_TUNE_CHAIN_ROWS_FWD = 7, // cells per workgroup of the forward chain launches: 0 = 64 (as the backward chains), > 0 = this
                                   // many (multiple of 8, <= 64; measured: no gain from smaller blocks, api.hip make_layout)
    MMVAE_TUNE_DW11_LDS = 9,       // dW11 beside the backward chain: KB of dynamic LDS added to its workgroups (40 fills the CU: kernels
                                   // that use any LDS -- the latent backward -- then stay off the CUs dW11 holds)
    MMVAE_TUNE_COUPLE_LATE = 10,   // fused step: the coupling kernel + T sums behind the dW11 fork (one fork fewer on the main stream)
                                   // instead of beside the decoder chain
    MMVAE_TUNE_JOIN_LAST = 11,     // fused step: the side stream is joined BEHIND the last reduction (which needs nothing from it)
    MMVAE_TUNE_FORK_RECORD = 12,   // fork events through hipEventRecord behind the kernel instead of riding on it (launch_k, common.hpp)
    MMVAE_TUNE_COUPLE_SIDE = 13,   // fused step, where the coupling terms run: 0 = as a role of the decoder chain's launch from four arms up
                                   // and on the side stream below (chain.hip dec_couple_ok), 1 = side stream always (fork behind the
                                   // latent forward, join in front of the latent backward), 3 = role always, 2 = timing experiment in round 3 (forcing the general-width kernels at fc_dim 100, fc11 grid shape, fc11 ablations)
    MMVAE_TUNE_FC11_ZG_OFF = 8,    // fc11 forward, loss and d(d10) as separate launches instead of the fused kernel
    MMVAE_TUNE_ABLATE_L = 14,      // latent kernels: ablations / stamps
    MMVAE_TUNE_LAT_FULLWAVE = 15,  // latent kernels: one wave per cell instead of the half-wave layout
    MMVAE_TUNE_ABLATE_B = 16,      // bf16 GEMM engine: 1 no MFMAs, 2 no global loads, 4 no LDS stores (results wrong)
    // 17 MMVAE_TUNE_ENGINE: public (mmvae.h)
    MMVAE_TUNE_BF16_NARROW_FP32 = 18, // bf16 configuration on bf16 storage: the narrow operands of fc1 / dW1 (W1, dZ1) read as fp32 and rounded
                                   // by every block tile, as before, instead of as bf16 from slice 0 of their planes
    MMVAE_TUNE_BN_PARTIALS = 19,   // BatchNorm batch sums through per-workgroup partial arrays instead of the accumulators
    MMVAE_TUNE_PRESPLIT_ALL = 20,  // fp32x3 engine: all slice planes through k_presplit launches
    MMVAE_TUNE_CHAIN_FP32 = 21,    // fp32x3 engine: the chain kernels' own GEMMs stay on the fp32 matrix instruction
    MMVAE_TUNE_REDUCE11_MAIN = 22, // fused Adam: reduce / update the fc11 tensors on the main stream with the rest
    MMVAE_TUNE_FUSED_CHAIN = 23,   // training mode, fc2..fc5 and their backward as ONE launch per chain with an in-launch barrier per
                                   // BatchNorm (chain.hip k_enc_fwd_fused / k_enc_bwd_fused): 1 = on (measured: the same step time
                                   // at A = 2, slower at A = 3: DESIGN.md section 15); tests: 2 = on and every third workgroup exits
                                   // at once (the others pick its row blocks up), 3 = on even when the grid exceeds the chip,
                                   // 4 / 5 = forward / backward chain only
};
