"""The frame loop of the reference's `run.py` (:203-433) as two functions:
`detect_frame` is one trip through the loop body -- KITTI files -> camera-frame
crop -> graph -> GNN -> softmax -> decode + NMS -> the 16-field KITTI rows --
with every stage on the device, and `run_dataset` is the loop with the
reference's checkpoint restore (:192-201), per-stage `time_dict` (:216-263,
:326, :412) and `<output_dir>/data/<frame>.txt` files (:414-423).

No session, placeholders, visualisation or command line: `config` is the
dictionary `util/config_util.load_config` returns for the reference's config
files, `dataset` a `pointgnn_amd.kitti_dataset.KittiDataset`.
"""
import os
import time

import torch

from . import graph_gen, kitti_output, models, nms, tf_bundle

BOX_ENCODING_LEN = 7


def build_model(config, checkpoint_dir=None, params=None, edge_arith='f32'):
    """run.py:135-141 + :192-201: the model in 'test' mode with the weights of
    the latest checkpoint under `checkpoint_dir` (or a ready name->array
    mapping).  `edge_arith` (extension): gnn.EDGE_ARITHS, default the fp32
    path."""
    model = models.get_model(config['model_name'])(
        num_classes=config['num_classes'], box_encoding_len=BOX_ENCODING_LEN,
        mode='test', **config['model_kwargs'])
    if params is None:
        params = tf_bundle.load_checkpoint(checkpoint_dir)
    params = {k: v for k, v in params.items() if k != 'Variable'}
    model.load_state_dict(params)
    model.edge_arith = edge_arith
    return model


def _input_features(config, points):
    """run.py:224-249: the `input_features` switch of the config."""
    kind = config['input_features']
    attr = points.attr
    if kind == 'irgb':
        return attr
    if kind == 'i':
        return attr[:, :1]
    if kind == '0':
        return torch.zeros_like(attr[:, :1])
    if kind == '0000':
        return torch.zeros_like(attr)
    if kind == '0rgb':
        out = attr.clone()
        out[:, 0] = 0
        return out
    if kind == 'i000':
        out = torch.zeros_like(attr)
        out[:, 0] = attr[:, 0]
        return out
    raise ValueError("input_features %r" % (kind,))


def detect_frame(dataset, frame_idx, model, config, use_box_merge=True,
                 use_box_score=True, time_dict=None, image_reader=None):
    """One iteration of run.py:203-412.  Returns (rows, stages): the KITTI
    rows of the frame and the device tensors of every stage (points, graph,
    logits, box encodings, probs, NMS outputs) for callers that check them."""
    td = time_dict if time_dict is not None else {}

    def lap(key, t_prev):
        torch.cuda.synchronize()
        now = time.time()
        td[key] = td.get(key, 0.0) + now - t_prev
        return now

    t = time.time()
    # run.py:210-211 (colour channels are zero unless `image_reader` supplies
    # the decoded BGR image: PNG decoding is not part of this package)
    image = image_reader(frame_idx) if image_reader is not None else None
    points = dataset.get_cam_points_in_image_with_rgb(
        frame_idx, config['downsample_by_voxel_size'], image=image)
    calib = dataset.get_calib(frame_idx)
    t = lap('fetch input', t)
    fn = graph_gen.get_graph_generate_fn(config['graph_gen_method'])
    coords, kps, edges = fn(points.xyz, **config['runtime_graph_gen_kwargs'])
    t = lap('gen graph', t)
    input_v = _input_features(config, points)
    logits, box_encodings = model.predict(input_v, coords, kps, edges, False)
    if model.edge_arith == 'f16x2' and not model.edge_range_ok():
        # an activation left fp16's safe range (gnn.EDGE_ARITHS): this frame
        # in fp32 (the results are read below anyway: no extra wait)
        td['f16x2 range reruns'] = td.get('f16x2 range reruns', 0) + 1
        model.edge_arith = 'f32'
        try:
            logits, box_encodings = model.predict(input_v, coords, kps, edges,
                                                  False)
        finally:
            model.edge_arith = 'f16x2'
    probs = model.postprocess(logits)
    t = lap('gnn inference', t)
    label_map = kitti_output.LABEL_MAPS[config['label_method']]
    labels, boxes, scores, nms_idx = nms.detect_boxes(
        probs, box_encodings, coords[-1], label_map,
        config['nms_overlapped_thres'],
        box_encoding_method=config['box_encoding_method'],
        use_box_merge=use_box_merge, use_box_score=use_box_score)
    cand_idx, _ = nms.select_candidates(probs)
    cand_xyz = coords[-1][(cand_idx // config['num_classes']).long()]
    t = lap('decode box + nms', t)
    rows = kitti_output.detections_to_kitti_labels(
        labels, boxes, scores, calib, config['label_method'],
        candidate_xyz=cand_xyz, use_box_score=use_box_score)
    lap('kitti rows', t)
    stages = {'points': points, 'calib': calib, 'coords': coords, 'kps': kps,
              'edges': edges, 'logits': logits, 'box_encodings': box_encodings,
              'probs': probs, 'class_labels': labels, 'boxes_3d': boxes,
              'scores': scores, 'nms_indices': nms_idx,
              'candidate_indices': cand_idx, 'candidate_xyz': cand_xyz}
    return rows, stages


def run_dataset(dataset, config, checkpoint_dir, output_dir,
                frame_indices=None, use_box_merge=True, use_box_score=True,
                params=None, log=None, image_reader=None, edge_arith='f32'):
    """run.py:203-433 over `frame_indices` (default: the whole dataset).
    Writes `<output_dir>/data/<frame name>.txt` in the reference's format and
    returns the accumulated `time_dict` (seconds per stage, plus 'frames')."""
    model = build_model(config, checkpoint_dir, params, edge_arith)
    if frame_indices is None:
        frame_indices = range(dataset.num_files)
    time_dict = {}
    n = 0
    for frame_idx in frame_indices:
        rows, _ = detect_frame(dataset, frame_idx, model, config,
                               use_box_merge, use_box_score, time_dict,
                               image_reader)
        t = time.time()
        kitti_output.write_kitti_txt(
            os.path.join(output_dir, 'data',
                         dataset.get_filename(frame_idx) + '.txt'), rows)
        time_dict['write txt'] = time_dict.get('write txt', 0.0) + \
            time.time() - t
        n += 1
        if log is not None:
            log("frame %s: %d detections" % (dataset.get_filename(frame_idx),
                                             len(rows)))
    time_dict['frames'] = n
    return time_dict
