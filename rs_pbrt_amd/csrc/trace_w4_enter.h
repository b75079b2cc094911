// k_trace_w4: TransformedPrimitive::intersect / intersect_p's first half (primitive.rs:216-232) — the world ray into the object's space, the object aggregate's root box.
// Textually included at its two sites (the leaf loop of the static-instance kernels, the entry phase of <.., ANIM>): as a lambda called from both, its by-reference
// captures put the lane's traversal state into scratch (536 B per lane; the C5 stand-in fell from 266 to 112 Msamples/s).  Reads and writes the kernel's lane state.
{
        const InstDev& in = sc.inst[inst];
        const float4* rp = reinterpret_cast<const float4*>(((entry & RSPT_Q_MIS) ? rays_b : rays_a) + (entry & ~RSPT_Q_MIS));
        const float4 r0 = rp[0], r1 = rp[1];
        f3 no, nd;
        w_tmax = t_max; sp_base = sp; inst_hit = false;
        if (ANIM && in.anim != RSPT_MISS) {   // primitive_to_world.interpolate(r.time) and its inverse (primitive.rs:218-222)
            const uint32_t slot = entry & ~RSPT_Q_MIS;
            const float time = (OUT_MODE == 0 && sc.ray_time) ? sc.ray_time[slot / sc.time_div] : 0.0f;
            float mi[12], mi3[4];   // (round 6: the inverse alone, inst_inverse_at — the traversal reads nothing else of the interpolated Transform)
            inst_inverse_at(sc, in, time, !sc.inst_fixed, mi, mi3, &inst_ident);
            xf_ray(mi, mi3, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, t_max, &no, &nd, &t_max);
        } else {
            inst_ident = in.identity != 0u;
            inst_ray(in, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, t_max, &no, &nd, &t_max);
        }
        ox = no.x; oy = no.y; oz = no.z;
        ix = 1.0f / nd.x; iy = 1.0f / nd.y; iz = 1.0f / nd.z;
        negbits = (ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u);
        if (!(fabsf(ix) < RSPT_INF && fabsf(iy) < RSPT_INF && fabsf(iz) < RSPT_INF)) negbits |= 8u;
        rs = ray_shear(nd);
        if (in.root_node == RSPT_MISS) leaf = in.w4_root;  // a lone GeometricPrimitive: no box test (api.rs:3046)
        else {                                            // the object aggregate's node 0 (bvh.rs:424)
            const float4 q0 = sc.nodes[2 * (size_t)in.root_node], q1 = sc.nodes[2 * (size_t)in.root_node + 1];
            if (box_hit(q0, q1, no, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max)) {
                if (in.w4_root & RSPT_REF_LEAF) leaf = in.w4_root;
                else cur = in.w4_root;
            }
        }
}
