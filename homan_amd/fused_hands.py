"""The fused iteration for TWO hands per frame (reference homan/homan.py:341-358, homan/lossutils.py:51-59,116-127,
homan/losses.py:207-241): the launch sequence `FusedStepper.forward_backward` issues when `hand_nb == 2`.  Split out of
homan_amd/fused.py; `self` is the FusedStepper (its buffers for this path are allocated in its constructor)."""
import torch

from . import lib as _lib


def forward_backward_hands(self, log=False):
    """The iteration for TWO hands per frame (reference homan.py:341-358, lossutils.py:51-59,116-127, losses.py:207-241),
    one clip.  Hand rows are interleaved frame-major [h0_t0, h1_t0, h0_t1, ...] like the model's Parameters: the MANO
    launches walk the strided slice of their hand through its side's model (hm_mano_*_rows), the hand-only terms and the
    rigid backward run once over all rows (the kernels' hand_nb), and the pair-wise terms see each hand as a dense copy:
    contact = mean over the hands, interaction = their sum, collision = the three two-mesh scenes (h0|h1), (h0|obj),
    (h1|obj), logged distance = largest per-frame distance to the NEAREST hand.  Same kernels and values as
    HOMan.forward + autograd; not tuned like the one-hand sequence (no fused pair-term launch)."""
    m, L, P, ck = self.model, self.L, _lib.ptr, _lib.check
    B, N, h, Vo, Vh, c, on, w = self.B, self.N, self.h, self.Vo, self.Vh, self.c, self.on, self.w
    NS = self.NS
    main, side = torch.cuda.current_stream(), self.side
    sa, sb = main.cuda_stream, side.cuda_stream
    rws_a, rws_b = P(m.reduce_ws.buf), P(self.reduce_ws_b.buf)
    sctx, cctx = m.sil_ctx, m.collision_ctx
    pca, rot, betas = m.mano_pca_pose, m.mano_rot, m.mano_betas
    mtr = m.mano_trans if m.optimize_mano else None
    slot = self._slot
    side.wait_stream(main)
    # ---------------- A: silhouettes forward + backward (as in the one-hand sequence)
    if on["sil"]:
        fwd_args = (P(m.verts_object_og), P(sctx.faces), 0, P(self.sil_K), B, Vo, sctx.F, sctx.S,
                    1.0, self.ops.NMR_NEAR, self.ops.NMR_FAR, P(self.sil_keep), P(self.sil_ref),
                    None, P(self.pooled), None, P(sctx.work_order), None, None, 0, P(m.rotations_object),
                    P(m.translations_object), P(m.int_scales_object), 1, 1, P(sctx.workspace), 0, NS, P(self.vo))
        ck(L.hm_sil_fwd_phase_clips(*fwd_args, 1, sa), "sil_fwd(setup)")
        self.ev_sil.record(main)
        ck(L.hm_sil_fwd_phase_clips(*fwd_args, 2, sa), "sil_fwd(raster)")
        ck(L.hm_sil_bwd_clips(P(self.vo), P(self.sil_K), B, Vo, sctx.F, sctx.S, 1.0, self.sil_eps,
                              2 if self.lw["lw_sil_obj"] > 0 else 1, P(self.up_sil), None, P(m.keep_sum),
                              P(sctx.adj_off), P(sctx.adj_items), P(sctx.face_order), None, None, P(sctx.workspace), 0,
                              slot("loss_sil_obj"), NS, sctx.sum_log2q, sa), "sil_bwd")
    # ---------------- B: hands
    with torch.cuda.stream(side):
        if not on["sil"]:
            ck(L.hm_rigid_fwd_clips(P(m.verts_object_og), P(m.rotations_object), P(m.translations_object),
                                    P(m.int_scales_object), 1, B, Vo, None, P(self.vo), 0, sb), "rigid_fwd(obj)")
            self.ev_vo.record(side)
        if m.optimize_mano:
            for i, hctx in enumerate(self.hand_ctx):       # hand i = rows i::h through the model of its side
                ck(L.hm_mano_fwd_rows(hctx.ptrs, P(pca), self.P, P(rot), P(betas), P(mtr), B, P(self.vm), None,
                                      P(m.rotations_hand), P(m.translations_hand), P(m.int_scales_hand), P(self.vh),
                                      P(self.mano_state), 0, i, h, sb), "mano_fwd + rigid(hand %d)" % i)
        else:
            ck(L.hm_rigid_fwd_clips(P(m.verts_hand_og), P(m.rotations_hand), P(m.translations_hand),
                                    P(m.int_scales_hand), 0, N, Vh, None, P(self.vh), 0, sb), "rigid_fwd(hands)")
        if on["pca"] or on["so"] or on["sh"]:
            ck(L.hm_priors_fwd_clips(P(pca), self.P * N, P(m.int_scales_object), P(m.int_scale_object_mean),
                                     P(m.int_scales_hand), P(m.int_scale_hand_mean), P(self.U_pca), P(self.U_so),
                                     P(self.U_sh), slot("loss_pca"), 1, NS, sb), "priors")
        if on["smooth"]:
            ck(L.hm_smooth_fwd_clips(P(self.vh), N, Vh, h, P(self.U_smh), slot("loss_smooth_hand"), rws_b, 0, NS, sb),
               "smooth(hands)")
        if on["v2d"]:
            ck(L.hm_v2d_fwd_clips(P(self.vh), P(m.camintr), h, P(m.ref_verts2d_hand), float(m.image_size), N, Vh,
                                  P(self.U_v2d), slot("loss_v2d_hand"), rws_b, 0, NS, sb), "v2d")
        if on["sil"]:
            side.wait_event(self.ev_sil)
        if on["smooth"]:
            ck(L.hm_smooth_fwd_clips(P(self.vo), B, Vo, 1, P(self.U_smo), slot("loss_smooth_obj"), rws_b, 0, NS, sb),
               "smooth(obj)")
        pairwise = on["con"] or on["inter"] or on["col"] or on["depth"]
        if pairwise:
            for i in range(h):
                self.vh_d[i].copy_(self.vh[i::h])
        tmp = self.tmp_h
        for i in range(h):
            rws_i = P(self.rws_h[i].buf)
            if on["con"] or on["inter"]:
                ck(L.hm_nn_fwd(P(self.vh_d[i]), P(self.vo), B, Vh, Vo, P(self.nn_idx_d[i]), P(self.nn_d2_d[i]),
                               tmp[i, 2:3].data_ptr(), rws_i, sb), "nn(hand %d)" % i)
            if on["con"]:
                ck(L.hm_contact_fwd(P(self.vh_d[i]), P(self.vo), P(self.nn_idx_d[i]), B, Vh, Vo, c.COLLISION_THRESH,
                                    P(self.U_conh_d[i]), P(self.U_cono_d[i]), tmp[i, 0:1].data_ptr(), rws_i, sb),
                   "contact(hand %d)" % i)
            if on["inter"]:
                ck(L.hm_inter_fwd(P(self.vh_d[i]), P(self.vo), P(m.camintr), B, Vh, Vo, c.INTERACTION_BBOX_EXPANSION,
                                  float(c.INTERACTION_Z_THRESH), P(self.rec_d[i]), tmp[i, 1:2].data_ptr(), rws_i, sb),
                   "inter(hand %d)" % i)
                if m.optimize_object_scale:
                    ck(L.hm_inter_bwd(P(self.rec_d[i]), P(self.up_inter), B, Vh, Vo, None, P(self.G_int_o_d[i]), sb),
                       "inter_bwd(hand %d)" % i)
        if on["col"]:
            # scene [hand 0, hand 1, object] (lossutils.py:53-59): the three two-mesh scenes of HOMan.collision_ctx
            scenes = ((self.vh_d[0], self.vh_d[1], self.U_col_d[0], self.U_col_d[1]),
                      (self.vh_d[0], self.vo, self.U_col_d[2], self.U_colo_d),
                      (self.vh_d[1], self.vo, self.U_col_d[3], self.U_colo_d))
            for k, (cc, (va, vb, ga, gb)) in enumerate(zip(cctx, scenes)):
                ck(L.hm_collision_fwd(P(va), P(cc.f0), cc.V0, cc.f0.shape[0], P(vb), P(cc.f1), cc.V1, cc.f1.shape[0], B,
                                      c.SDF_SCALE_FACTOR, P(ga), P(gb), self.tmp_col[k:k + 1].data_ptr(), P(cc.ws), sb),
                   "collision(scene %d)" % k)
        if on["depth"]:
            # three depth renders at the full-image camera, the three pairs' ordinal terms, the scene's normaliser and
            # every pair's share of it on the device, the pairs' backward passes with that share (times the weight) as
            # upstream, a layer's two gradient images added, one depth-map backward per layer
            Sd, K = self.dlayers[0][0].S, P(m.camintr)
            lverts = [self.vo] + list(self.vh_d)
            for li, (ctx, V_, _) in enumerate(self.dlayers):
                self._depth_render(lverts[li], ctx, V_, self.dl_sil[li], self.dl_dep[li], sb)
            for k, (a, b) in enumerate(self.dpairs):
                ck(L.hm_ordinal_depth_fwd(P(self.dl_dep[a]), P(self.dl_dep[b]), P(self.dl_sil[a]), P(self.dl_sil[b]),
                                          P(self.dlayers[a][2]), P(self.dlayers[b][2]), B, Sd, P(self.dp_part[k]),
                                          P(self.dp_rec[k]), P(self.dp_out[k]), P(self.rws_dp[k].buf), sb), "ordinal depth")
            present = [(sl == 1).flatten(1).any(1).sum().float() for sl in self.dl_sil]
            npairs = [self.dp_rec[k][0] for k in range(len(self.dpairs))]
            total = sum(present) + sum(npairs[k] - present[a] - present[b] for k, (a, b) in enumerate(self.dpairs))
            loss = torch.zeros((), device=self.vo.device)
            for k in range(len(self.dpairs)):
                share = torch.where(npairs[k] > 0, npairs[k] / total, torch.zeros_like(total))
                loss = loss + torch.where(npairs[k] > 0, self.dp_out[k][0] * share, torch.zeros_like(total))
                self.dp_up[k].copy_((w["loss_depth"] * share).reshape(1))
            self.vals[0][self.SLOTS.index("loss_depth")] = loss
            for li in range(len(self.dlayers)):
                self.dl_g[li].zero_()
            for k, (a, b) in enumerate(self.dpairs):
                ga, gb = self.dp_g[k]
                ck(L.hm_ordinal_depth_bwd(P(self.dl_dep[a]), P(self.dl_dep[b]), P(self.dl_sil[a]), P(self.dl_sil[b]),
                                          P(self.dlayers[a][2]), P(self.dlayers[b][2]), B, Sd, P(self.dp_rec[k]),
                                          P(self.dp_up[k]), P(ga), P(gb), sb), "ordinal depth bwd")
                self.dl_g[a].add_(ga)
                self.dl_g[b].add_(gb)
            gouts = [self.G_dep_o] + list(self.G_dep_h_d)
            for li, (ctx, V_, _) in enumerate(self.dlayers):
                ck(L.hm_depth_bwd(P(lverts[li]), K, B, V_, ctx.F, Sd, 1.0, P(self.dl_g[li]), P(ctx.adj_off),
                                  P(ctx.adj_items), P(gouts[li]), P(ctx.workspace), sb), "depth bwd")
            for i in range(h):
                self.G_dep_h[i::h].copy_(self.G_dep_h_d[i])
        # ---- the hands' values combined like the reference does, gradients back onto the interleaved rows
        with torch.cuda.stream(side):
            v0 = self.vals[0]
            if on["con"]:
                v0[self.SLOTS.index("loss_contact")] = torch.stack([tmp[i, 0] for i in range(h)]).mean()
                for i in range(h):
                    self.U_conh[i::h].copy_(self.U_conh_d[i])
            if on["inter"]:
                v0[self.SLOTS.index("loss_inter")] = tmp[0, 1] + tmp[1, 1]
                per_frame = torch.stack([d.min(1)[0] for d in self.nn_d2_d]).min(0)[0]
                v0[self.SLOTS.index("handobj_maxdist")] = per_frame.max().clamp_min(0).sqrt()
                for i in range(h):
                    self.rec[i::h].copy_(self.rec_d[i])
                if m.optimize_object_scale:
                    torch.add(self.G_int_o_d[0], self.G_int_o_d[1], out=self.G_int_o)
            if on["col"]:
                v0[self.SLOTS.index("loss_collision")] = (self.tmp_col[0] + self.tmp_col[1]) + self.tmp_col[2]
                self.U_colh[0::h].copy_(self.U_col_d[0])      # from the (h0|h1) scene ...
                self.U_colh[1::h].copy_(self.U_col_d[1])
                self.U_colh2[0::h].copy_(self.U_col_d[2])     # ... and from each hand's scene with the object
                self.U_colh2[1::h].copy_(self.U_col_d[3])
        self.ev_pair.record(side)
        if on["depth"] and on["col"]:
            # (a rigid backward sums five weighted terms: with the depth term the two collision buffers share a slot -
            #  another float summation order than without it, and nothing is written out for this combination)
            self.U_colh.add_(self.U_colh2)
        tp, tw, tn = _lib.terms([(self.U_smh if on["smooth"] else None, w["loss_smooth_hand"]),
                                 (self.U_v2d if on["v2d"] else None, w["loss_v2d_hand"]),
                                 (self.U_colh if on["col"] else None, w["loss_collision"]),
                                 ((self.G_dep_h, 1.0) if on["depth"] else
                                  (self.U_colh2 if on["col"] else None, w["loss_collision"])),
                                 (self.U_conh if on["con"] else None, w["loss_contact"] / h)])
        ck(L.hm_rigid_bwd_clips(P(self.vm if m.optimize_mano else m.verts_hand_og), P(m.rotations_hand),
                                P(m.int_scales_hand), 0, tp, tw, tn, None,
                                (self.rec.data_ptr() + 8) if on["inter"] else None, 8, w["loss_inter"] / Vh, N, Vh,
                                P(self.G_mesh) if m.optimize_mano else None, P(m.rotations_hand.grad),
                                P(m.translations_hand.grad), None, P(self.rigid_ws_h), 0, sb), "rigid_bwd(hands)")
        if m.optimize_mano:
            for i, hctx in enumerate(self.hand_ctx):
                ck(L.hm_mano_bwd_rows(hctx.ptrs, P(pca), self.P, P(rot), P(betas), B, P(self.G_mesh),
                                      P(self.U_pca) if on["pca"] else None, w["loss_pca"], P(pca.grad), P(rot.grad),
                                      P(betas.grad), P(mtr.grad), P(self.mano_state), P(hctx.workspace(B)), i, h, sb),
                   "mano_bwd(hand %d)" % i)
    # ---------------- A: object backward
    main.wait_event(self.ev_pair)
    sc_obj = m.optimize_object_scale
    tp, tw, tn = _lib.terms([(self.U_smo if on["smooth"] else None, w["loss_smooth_obj"]),
                             (self.U_cono_d[0] if on["con"] else None, w["loss_contact"] / h),
                             (self.U_cono_d[1] if on["con"] else None, w["loss_contact"] / h),
                             (self.G_int_o if (on["inter"] and sc_obj) else None, 1.0),
                             (self.G_dep_o if on["depth"] else None, 1.0)])
    if on["sil"]:
        ck(L.hm_rigid_bwd_sil_clips(P(m.verts_object_og), P(m.rotations_object), P(m.int_scales_object), 1, tp, tw, tn,
                                    L.hm_sil_parts(P(sctx.workspace), B, Vo, sctx.F, sctx.S), P(sctx.adj_off),
                                    P(sctx.adj_items), P(self.vo), P(self.sil_K), 1.0, sctx.F, B, Vo,
                                    P(m.rotations_object.grad), P(m.translations_object.grad),
                                    P(self.g_so_part) if sc_obj else None, P(self.rigid_ws_o), 0, sctx.sum_log2q, None, 0.0, sa),
           "rigid_bwd(obj) + silhouette gather")
    else:
        ck(L.hm_rigid_bwd_clips(P(m.verts_object_og), P(m.rotations_object), P(m.int_scales_object), 1, tp, tw, tn, None,
                                None, 0, 0.0, B, Vo, None, P(m.rotations_object.grad), P(m.translations_object.grad),
                                P(self.g_so_part) if sc_obj else None, P(self.rigid_ws_o), 0, sa), "rigid_bwd(obj)")
    main.wait_stream(side)
    if log:
        ck(L.hm_log_total_clips(P(self.vals), P(self.weights), len(self.SLOTS), P(self.opt.step_t), self.max_steps,
                                P(self.log_buf), 1, sa), "log")
    if sc_obj:
        ck(L.hm_sum_small_clips(P(self.g_so_part), B, 1.0, P(self.U_so) if on["so"] else None, w["loss_scale_obj"],
                                P(m.int_scales_object.grad), 1, sa), "scale grad")
