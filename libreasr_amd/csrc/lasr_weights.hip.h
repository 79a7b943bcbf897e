// lasr_weights.hip.h -- weight blob reader: BatchNorm fold, LSTM layer packing
// Engine unit only (lasr_engine.hip), included after lasr_decode.hip.h.
#pragma once

namespace {

// ---------------------------------------------------------------------------- weight loading
struct Reader {
    const float* p; size_t left;
    const float* take(size_t n) {
        if (n > left) return nullptr;
        const float* q = p; p += n; left -= n; return q;
    }
};

int fold_bn(lasr_ctx* c, Reader& rd, int H, float** s_dev, float** t_dev) {
    const float* w = rd.take(H); const float* b = rd.take(H); const float* mean = rd.take(H); const float* var = rd.take(H);
    if (!var) return fail(c, LASR_EINVAL, "weight blob too short (bn)");
    std::vector<float> s(H), t(H);
    for (int i = 0; i < H; ++i) {
        const double sc = (double)w[i] / std::sqrt((double)var[i] + 1e-5);   // BatchNorm1d eps
        s[i] = (float)sc;
        t[i] = (float)((double)b[i] - (double)mean[i] * sc);
    }
    RC(upload(c, s_dev, s.data(), H));
    RC(upload(c, t_dev, t.data(), H));
    return LASR_OK;
}

// LSTM layer in torch layout: W_ih [4H,I], W_hh [4H,H].
//   tiling C (encoder): tile t = (jb = t/2, nt = t%2): column col -> gate 2*nt + col/8, unit 8*jb + col%8
//   tiling A (predictor): tile jb = 4 units x 4 gates (col = gate*4 + unit)
//   tiling D (encoder, c->enc_u12): tile t = (jb = t/3, nt = t%3): column c48 = 16*nt + col -> gate c48/12, unit 12*jb + c48%12
int load_lstm(lasr_ctx* c, Reader& rd, Cell& L, int I, int H, bool tiling_a, std::vector<float>* keep_wih,
              std::vector<float>* keep_bias) {
    L.I = I;
    const float* wih = rd.take((size_t)4 * H * I); const float* whh = rd.take((size_t)4 * H * H);
    const float* bih = rd.take(4 * H); const float* bhh = rd.take(4 * H);
    if (!bhh) return fail(c, LASR_EINVAL, "weight blob too short (lstm)");
    Packed pk;
    if (!tiling_a && c->enc_u12) {          // (packed into the WxC / WhC slots: one encoder tiling per context)
        auto src_row = [&](int t, int col) { const int c48 = 16 * (t % 3) + col; return (size_t)(c48 / 12) * H + 12 * (t / 3) + c48 % 12; };
        pack_tiles(pk, c->bf, (H / 12) * 3, I, [&](int t, int col, int k) { return wih[src_row(t, col) * I + k]; });
        RC(upload_packed(c, &L.WxC, pk));
        pack_tiles(pk, c->bf, (H / 12) * 3, H, [&](int t, int col, int k) { return whh[src_row(t, col) * H + k]; });
        RC(upload_packed(c, &L.WhC, pk));
    } else
    if (!tiling_a) {
        pack_tiles(pk, c->bf, (H / 8) * 2, I, [&](int t, int col, int k) { return wih[((size_t)(2 * (t & 1) + (col >> 3)) * H + 8 * (t >> 1) + (col & 7)) * I + k]; });
        RC(upload_packed(c, &L.WxC, pk));
        pack_tiles(pk, c->bf, (H / 8) * 2, H, [&](int t, int col, int k) { return whh[((size_t)(2 * (t & 1) + (col >> 3)) * H + 8 * (t >> 1) + (col & 7)) * H + k]; });
        RC(upload_packed(c, &L.WhC, pk));
    } else {
        pack_tiles(pk, c->bf, H / 4, I, [&](int t, int col, int k) { return wih[((size_t)(col >> 2) * H + 4 * t + (col & 3)) * I + k]; });
        RC(upload_packed(c, &L.WxA, pk));
        pack_tiles(pk, c->bf, H / 4, H, [&](int t, int col, int k) { return whh[((size_t)(col >> 2) * H + 4 * t + (col & 3)) * H + k]; });
        RC(upload_packed(c, &L.WhA, pk));
    }
    std::vector<float> bias(4 * H);
    for (int i = 0; i < 4 * H; ++i) bias[i] = bih[i] + bhh[i];
    RC(upload(c, &L.bias, bias.data(), bias.size()));
    if (keep_wih) keep_wih->assign(wih, wih + (size_t)4 * H * I);
    if (keep_bias) *keep_bias = bias;
    return LASR_OK;
}


}  // namespace
