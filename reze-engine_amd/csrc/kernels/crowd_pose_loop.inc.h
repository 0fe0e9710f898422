// kernels/crowd_pose_loop.inc.h — the pose loop of the crowd kernels, included TEXTUALLY into the body of rz_skin_instances_kernel and
// rz_skin_instances_fk_kernel (crowd.hip) right behind the barrier that publishes the workgroup's palettes. A device function with the
// same body compiled to a different instruction stream (22 instructions more, other schedule: tools/archive/isa_diff.py), and the C4 frame sits
// on the store stream's edge — so the two kernels share the text, not a call.
// In scope at the point of inclusion: BLOCK, NTS, SUB (compile time); p, pal, ng, lrows, inst0, Vp, v_begin, v_end, bmax, jp01, jp23,
// tid, vert_of(); the run's FIRST vertex already loaded into v, x, y, z, nx, ny, nz, j01, j23, wq; rstride (float4 per palette bone).
// Software-pipelined: the next vertex's nine attribute loads are issued before the current vertex's pose loop, so their L2 latency
// hides behind the poses' LDS gathers + FMA.
    for (uint32_t vb = v_begin; vb < v_end; vb += BLOCK) {   // workgroup-uniform trip count (the ballots below need whole waves)
        if (vb == v_begin + BLOCK) RZ_STAMP(3);      // first vertex step done (8 poses written)
        const uint32_t vn = vert_of(vb + BLOCK);
        float xn = 0, yn = 0, zn = 0, nxn = 0, nyn = 0, nzn = 0;
        uint32_t j01n = 0, j23n = 0, wqn = 0;
        if (vn < v_end) {
            xn = p.geom[0 * Vp + vn]; yn = p.geom[1 * Vp + vn]; zn = p.geom[2 * Vp + vn];
            nxn = p.geom[3 * Vp + vn]; nyn = p.geom[4 * Vp + vn]; nzn = p.geom[5 * Vp + vn];
            j01n = jp01[vn]; j23n = jp23[vn]; wqn = p.weights[vn];
        }
        const bool live = v < v_end;
        // decode once per vertex (engine.ts:255-258)
        const uint32_t b0 = wq & 255u, b1 = (wq >> 8) & 255u, b2 = (wq >> 16) & 255u, b3 = wq >> 24;
        const uint32_t isum = b0 + b1 + b2 + b3;
        const bool ok = isum != 0u;
        const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
        const float w0 = ok ? (float)b0 * inv : 1.0f, w1 = (float)b1 * inv, w2 = (float)b2 * inv, w3 = (float)b3 * inv;
        const uint32_t jmax = SUB ? 0xffffu : bmax;     // SUB: slots are in range by construction
        const uint32_t o0 = min(j01 & 0xffffu, jmax) * rstride, o1 = min(j01 >> 16, jmax) * rstride,
                       o2 = min(j23 & 0xffffu, jmax) * rstride, o3 = min(j23 >> 16, jmax) * rstride;
        float *dp = p.out_pos + ((size_t)inst0 * Vp + v) * 3;
        float *dn = p.out_nrm + ((size_t)inst0 * Vp + v) * 3;
        // Packed-math form (v_pk_fma_f32 = two f32 FMAs per lane per instruction): palette rows are blended as
        // (xy),(zw) register pairs straight out of ds_read_b128, and position + normal are transformed together
        // as the pairs (x,nx),(y,ny),(z,nz), so one FMA chain yields (p_r, n_r) for row r. Every chain is spelled out
        // with explicit FMAs in the order of skin_vertex() above — bones ascending from w0 * row, then
        // fma(m.z, z, fma(m.y, y, fma(m.x, x, m.w))) — so a pose of a crowd has the SAME BITS as that pose run alone
        // through rz_deform_kernel (tests/test_gpu_round2.py checks it at full C4 size).
        const f2 vx = {x, nx}, vy = {y, ny}, vz = {z, nz};
        const f2 W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3};
#ifdef RZ_ABLATE
        if (p.dbg == 2 || p.dbg == 7) {          // dbg 2 / 7: ablation — the output stream without gathers / math
            if (live)
                for (int g = 0; g < ng; ++g) {
                    st3<NTS>(dp, x, y, z); st3<NTS>(dn, nx, ny, nz);
                    dp += Vp * 3; dn += Vp * 3;
                }
            x = xn; y = yn; z = zn; nx = nxn; ny = nyn; nz = nzn; j01 = j01n; j23 = j23n; wq = wqn;
            v = vn;
            continue;
        }
#endif
        auto pose_loop = [&](auto nb_tag) {
            constexpr int NB = decltype(nb_tag)::value;
            const float4 *pg = pal;
#pragma unroll 2
            for (int g = 0; g < ng; ++g) {
                f2 r[3][2];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 a = pg[o0 + k];
                    r[k][0] = W0 * f2{a.x, a.y};
                    r[k][1] = W0 * f2{a.z, a.w};
                    if (NB >= 2) {
                        const float4 c = pg[o1 + k];
                        r[k][0] = pk_fma(W1, f2{c.x, c.y}, r[k][0]);
                        r[k][1] = pk_fma(W1, f2{c.z, c.w}, r[k][1]);
                    }
                    if (NB >= 4) {
                        const float4 d = pg[o2 + k], e = pg[o3 + k];
                        r[k][0] = pk_fma(W3, f2{e.x, e.y}, pk_fma(W2, f2{d.x, d.y}, r[k][0]));
                        r[k][1] = pk_fma(W3, f2{e.z, e.w}, pk_fma(W2, f2{d.z, d.w}, r[k][1]));
                    }
                }
                // (p_r, t_r) = fma(m_r.z, (z,nz), fma(m_r.y, (y,ny), fma(m_r.x, (x,nx), (m_r.w, 0))))
                f2 q[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f2 m_xy = r[k][0], m_zw = r[k][1];
                    q[k] = pk_fma(f2{m_zw.x, m_zw.x}, vz, pk_fma(f2{m_xy.y, m_xy.y}, vy, pk_fma(f2{m_xy.x, m_xy.x}, vx, f2{m_zw.y, 0.0f})));
                }
                const float tx = q[0].y, ty = q[1].y, tz = q[2].y;
                const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
                const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
                if (live && (RZ_DBG(p) != 1 || l2 == 1234.5f)) {   // dbg 1 (tools-only build): compute without the output stream
                    st3<NTS>(dp, q[0].x, q[1].x, q[2].x);
                    st3<NTS>(dn, good ? tx * rl : nx, good ? ty * rl : ny, good ? tz * rl : nz);
                }
                pg += lrows;
                dp += Vp * 3;
                dn += Vp * 3;
            }
        };
        const bool any34 = __ballot((wq >> 16) != 0u) != 0ull;
        const bool any2 = __ballot(((wq >> 8) & 255u) != 0u) != 0ull;
        if (__ballot(live) == 0ull) {}                     // a wave past the end of the run (last step only)
        else if (any34) pose_loop(std::integral_constant<int, 4>{});
        else if (any2) pose_loop(std::integral_constant<int, 2>{});
        else pose_loop(std::integral_constant<int, 1>{});
        x = xn; y = yn; z = zn; nx = nxn; ny = nyn; nz = nzn; j01 = j01n; j23 = j23n; wq = wqn;
        v = vn;
    }
