# PanopticTrack inference configuration for vps_amd — same model/test_cfg keys and values as the reference's
# configs/cityscapes/track.py:2-86,132-148 (the plugin surface); training-only sections are omitted because this
# package implements the inference path. The reference's own config file also loads unchanged (extra keys are ignored
# or, for losses, accepted and unused).
num_things, num_classes = 8, 19
model = dict(
    type='PanopticTrack',
    pretrained=None,
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
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
    class_mapping={1: 11, 2: 12, 3: 13, 4: 14, 5: 15, 6: 16, 7: 17, 8: 18})
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
