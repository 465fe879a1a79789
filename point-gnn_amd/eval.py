"""`eval.py`'s evaluation pass (:68-132 fetch_data, :297-376 eval_once) as a
function: every labelled frame of the dataset through the model in 'eval'
mode, `model.loss` and the streaming metrics, with the reference's report.
The polling loop around it (`eval_repeat`, :377-397: wait for a new checkpoint,
sleep) is control plane and stays with the caller."""
import time

import torch

from . import metrics as metrics_mod
from . import models, preprocess, tf_bundle
from .train import fetch_data

BOX_ENCODING_LEN = 7


def eval_once(dataset, config, eval_config, checkpoint_dir=None, params=None,
              log=None):
    """eval.py:297-376.  `checkpoint_dir` (or a ready name->array mapping in
    `params`) supplies the weights; returns the final `results` dictionary
    (the running means / recall / precision / mAP after the last frame, plus
    'step' = the checkpoint's global step)."""
    if params is None:
        params = tf_bundle.load_checkpoint(checkpoint_dir)
    step = int(params['Variable']) if 'Variable' in params else 0
    params = {k: v for k, v in params.items() if k != 'Variable'}
    model = models.get_model(config['model_name'])(
        num_classes=config['num_classes'], box_encoding_len=BOX_ENCODING_LEN,
        mode='eval', **config['model_kwargs']).load_state_dict(params)
    n_samples = eval_config.get('NUM_TEST_SAMPLE', -1)
    if n_samples is None or n_samples < 0:
        n_samples = dataset.num_files
    aug_fn = preprocess.get_data_aug(eval_config.get('data_aug_configs', []))
    meter = metrics_mod.StreamingMetrics(config['num_classes'])
    start = time.time()
    results = {}

    def report(header):
        if log is None:
            return
        log(header)
        log('cls:%f, loc:%f, reg:%f, loss: %f' % (
            results['cls_loss'], results['loc_loss'], results['reg_loss'],
            results['total_loss']))
        log(meter.format(results))

    for frame_idx in range(n_samples):
        (input_v, coords, kps, edges, cls_labels, encoded, valid) = \
            fetch_data(dataset, frame_idx, config, eval_config, aug_fn)
        logits, pred_box = model.predict(input_v, coords, kps, edges,
                                         config.get('eval_is_training', True))
        probs = model.postprocess(logits)
        loss = model.loss(logits, cls_labels, pred_box, encoded, valid,
                          **config['loss'])
        results = meter.update(probs, cls_labels, loss)
        results['step'] = step
        if n_samples >= 10 and (frame_idx + 1) % (n_samples // 10) == 0:
            report('@frame %d' % frame_idx)
    if results:
        report('STEP: %d, time cost: %f' % (step, time.time() - start))
    return results
