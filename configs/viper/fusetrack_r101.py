# BASELINE config 5: the FuseTrack model with the ResNet-101 backbone variant (mmdet/models/backbones/resnet.py:361,
# arch_settings[101] = Bottleneck (3, 4, 23, 3)) for VIPER-scale 1080x1920 frames (Pad(32) -> 1088x1920 = 17*64 x 30*64, so
# FlowNet2's /64 rule holds without extra padding, panoptic_fusetrack.py:130). The reference ships no VIPER config (only
# tools/dataset/viper.py, the evaluation side); everything except `depth` is configs/cityscapes/fusetrack.py.
num_things, num_classes = 8, 19
model = dict(
    type='PanopticFuseTrack',
    pretrained=None,
    backbone=dict(type='ResNet', depth=101, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
    extra_neck=dict(type='BFPTcea', in_channels=256, num_levels=5, refine_level=0, refine_type='conv', center=0, nframes=2),
    panoptic=dict(type='UPSNetFPN', in_channels=256, out_channels=128, num_levels=4, num_things_classes=num_things,
                  num_classes=num_classes, ignore_label=255, loss_weight=1.0),
    rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256, anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                  anchor_strides=[4, 8, 16, 32, 64], target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                  loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0)),
    bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=2),
                            out_channels=256, featmap_strides=[4, 8, 16, 32]),
    bbox_head=dict(type='SharedFCBBoxHead', num_fcs=2, in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                   num_classes=num_things + 1, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2],
                   reg_class_agnostic=False),
    track_head=dict(type='TrackHead', num_fcs=2, in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                    match_coeff=[1.0, 2.0, 10.0]),
    mask_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=14, sample_num=2),
                            out_channels=256, featmap_strides=[4, 8, 16, 32]),
    mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256, num_classes=num_things + 1))
train_cfg = None
test_cfg = dict(
    rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000, nms_thr=0.7, min_bbox_size=0),
    rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_thr=0.5), max_per_img=100, mask_thr_binary=0.5),
    loss_pano_weight=None,
    flownet2=[],
    class_mapping={1: 11, 2: 12, 3: 13, 4: 14, 5: 15, 6: 16, 7: 17, 8: 18})
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
