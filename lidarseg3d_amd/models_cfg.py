"""Model config dicts of the reference's shipped configs, restated as plain dicts (only the `model` part):
configs/semanticnusc/SDSeg3D/semnusc_transvfe_unetscn3d_batchloss_e48.py:17-55 and
configs/semanticnusc/MSeg3D/semnusc_avgvfe_unetscn3d_hrnetw18_lr1en2_e12.py:57-123 (camera CNN omitted: its
outputs are inputs of the hot path).  Used by bench.py / tests; real config files load through
lidarseg3d_amd.config.Config.fromfile."""


def sdseg3d(num_class=17, cp=5, pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), voxel_size=(0.1, 0.1, 0.2)):
    return dict(
        type="SegNet", pretrained=None,
        reader=dict(type="TransformerVoxelFeatureExtractor", num_input_features=cp, num_compressed_features=16,
                    num_embed=64, num_head=4, num_layers=3),
        backbone=dict(type="UNetSCN3D", num_input_features=16, ds_factor=8, us_factor=8,
                      point_cloud_range=list(pc_range), voxel_size=list(voxel_size), model_cfg=dict(SCALING_RATIO=2)),
        point_head=dict(type="PointSegBatchlossHead", class_agnostic=False, num_class=num_class,
                        model_cfg=dict(CONV_IN_DIM=32, CONV_CLS_FC=[64], CONV_ALIGN_DIM=64, OUT_CLS_FC=[64, 64],
                                       IGNORED_LABEL=0)),
        voxel_generator=dict(range=list(pc_range), voxel_size=list(voxel_size), max_points_in_voxel=5,
                             max_voxel_num=[300000, 300000]),
    )


def mseg3d(num_class=17, cp=5, c_img=48, pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), voxel_size=(0.1, 0.1, 0.2)):
    return dict(
        type="SegMSeg3DNet", pretrained=None,
        reader=dict(type="ImprovedMeanVoxelFeatureExtractor", num_input_features=cp),
        backbone=dict(type="UNetSCN3D", num_input_features=cp + 8, ds_factor=8, us_factor=8,
                      point_cloud_range=list(pc_range), voxel_size=list(voxel_size), model_cfg=dict(SCALING_RATIO=2)),
        point_head=dict(type="PointSegMSeg3DHead", class_agnostic=False, num_class=num_class, model_cfg=dict(
            VOXEL_IN_DIM=32, VOXEL_CLS_FC=[64], VOXEL_ALIGN_DIM=64, IMAGE_IN_DIM=c_img, IMAGE_ALIGN_DIM=64,
            GEO_FUSED_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0, DP_RATIO=0.25, MIMIC_FC=[64, 64],
            SFPhase_CFG=dict(embeddings_proj_kernel_size=1, d_model=96, n_head=4, n_layer=6, n_ffn=192, drop_ratio=0,
                             activation="relu", pre_norm=False))),
        voxel_generator=dict(range=list(pc_range), voxel_size=list(voxel_size), max_points_in_voxel=5,
                             max_voxel_num=[300000, 300000]),
    )
