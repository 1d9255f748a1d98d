"""Module-as-singleton configuration with the reference's variable names and validation rules.

Mirrors /root/reference/configs/config.py: the module's globals ARE the configuration (config.py:8-63), a user
file can overlay them (`import_configs`, config.py:208-263), unknown names are rejected (config.py:242-243) and
`validate_configs` applies the same PGGAN checks and derives `transit_sch` from `transit_period`
(config.py:165-200).  Differences, all outside the training-step hot path: no interactive prompts (a clash
raises instead of asking, config.py:137-146) and directories are only created when `create_dirs=True`.
"""
import importlib.util
import os
import sys
import uuid
from types import FunctionType, ModuleType

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))

_DEFAULTS = dict(
    # directories
    root_dir=os.path.abspath(os.path.join(_HERE, os.pardir)), configs_dir=_HERE,
    # WGAN
    wgan=False, n_critic=1, adapt_critic=False, weights_init='', unroll_steps=0,
    # PGGAN
    pggan=True, grad_pen_lambda=10, transit_sch=[25000, 50000, 75000, 100000, 125000], transit_period=None,
    alpha_step=0.0001,
    # training
    ID=uuid.uuid4().hex[:4], RMSprop=False, learning_rate=0.0001, batch_size=8, N_epochs=150000, N_epochs_session=None,
    beta1=0.5, sim_loss_lambda=0.0, sim_loss_lambda_decay_rate=0.0, drift_epsilon=0.001, resume=False, N_workers=2,
    seed=1, checkpointing_period=100, device='default', pin_memory=False,
    # dataset
    dataset_name='science_2022', translation=0.05, image_preprocessing='cpu',
    # architecture
    latent_dim=512, image_size=512, N_colors=1, LeakyReLU_leak=0.2,
    N_gen_features=[128, 64, 32, 32, 16, 16], N_dis_features=[16, 16, 32, 32, 64, 128],
)
_DEFAULTS.update(
    data_dir=os.path.join(_DEFAULTS['root_dir'], 'data'), images_dir=os.path.join(_DEFAULTS['root_dir'], 'images'),
    weights_dir=os.path.join(_DEFAULTS['root_dir'], 'weights'), plots_dir=os.path.join(_DEFAULTS['root_dir'], 'plots'),
    logs_dir=os.path.join(_DEFAULTS['root_dir'], 'logs'))
_DEFAULTS.update(dataset_dir=os.path.join(_DEFAULTS['data_dir'], _DEFAULTS['dataset_name']),
                 samples_sub_dir=os.path.join(_DEFAULTS['images_dir'], _DEFAULTS['ID']))

globals().update(_DEFAULTS)
configs_name = dict(_DEFAULTS)  # the set of legal configuration names (reference: config.py:81)

# widths keyed by training ID (reference config.py:84-105)
_ID_WIDTHS = {
    ('0004', '0005'): ([1024, 512, 256, 128, 64, 32, 16, 8], [16, 32, 64, 128, 128, 128, 128]),
    ('0006',): ([512, 256, 128, 64, 32, 16, 8, 8], [64, 128, 256, 256, 256, 128, 64]),
    ('0007',): ([512, 256, 128, 64, 32, 16], [16, 32, 64, 128, 256, 512]),
    ('0008',): ([512, 256, 128, 64], [64, 128, 256, 512]),
    ('0009',): ([32, 32, 32, 32, 16, 16], [16, 16, 32, 32, 32, 32]),
    ('0010', '0011', '0012'): ([128, 64, 32, 32, 16, 16], [16, 16, 32, 32, 64, 128]),
}


def define_ID_dependent_configs():
    g = globals()
    assert g['ID'] != '', 'ID is not defined.'
    for ids, (gen, dis) in _ID_WIDTHS.items():
        if g['ID'] in ids:
            g['N_gen_features'], g['N_dis_features'] = list(gen), list(dis)
    g['samples_sub_dir'] = os.path.join(g['images_dir'], '{}'.format(g['ID']))


def print_configs():
    print('Configurations:')
    for name in configs_name:
        print(f'{name}:', globals()[name])


def validate_configs(create_dirs=False):
    g = globals()
    for d in ('dataset_dir', 'images_dir', 'samples_sub_dir', 'weights_dir', 'plots_dir'):
        g[d] = os.path.abspath(g[d])
    if create_dirs:
        for d in ('images_dir', 'weights_dir', 'plots_dir', 'logs_dir', 'samples_sub_dir'):
            os.makedirs(g[d], exist_ok=True)
    if g['device'] == 'default':
        g['device'] = 'cuda' if torch.cuda.is_available() else 'cpu'

    image_size_log = np.round(np.log2(g['image_size']))
    assert g['image_size'] == 2 ** image_size_log, 'Image size must be a power of 2.'
    assert g['device'] in ['cpu', 'cuda', 'mps'], f"device:{g['device']} is not supported."
    assert g['ID'] != '', 'The training ID is undefined.'
    if g['pggan']:
        err_msg = 'The number of layers in the generator and discriminator must match.'
        assert len(g['N_gen_features']) == len(g['N_dis_features']), err_msg
        N_upsamples = len(g['N_gen_features']) - 1
        assert g['image_size'] // (2 ** N_upsamples) >= 4, 'The initial image size must be >= 4. Reduce the number of layers'
        if g['transit_period'] is not None:
            g['transit_sch'] = [i * g['transit_period'] for i in range(1, N_upsamples + 1)]
        err_msg = 'The number of transitions ({}) does not match the number of convolution layers ({})'.format(
            len(g['transit_sch']), N_upsamples)
        assert N_upsamples == len(g['transit_sch']), err_msg
        assert g['N_epochs'] > g['transit_sch'][-1], 'The number of epochs must be greater than the last resolution transition'
        N_transition_epochs = np.ceil(1 / g['alpha_step'])
        err_msg = 'The transitions must be separated by at least {} epochs'.format(N_transition_epochs)
        assert np.all(np.diff(g['transit_sch']) > N_transition_epochs), err_msg


def set_configs(**overrides):
    """Programmatic overlay (what train.py does with CLI flags, train.py:101-104)."""
    for name, val in overrides.items():
        if name not in configs_name:
            raise ValueError(f"The overwritten config '{name}' is not defined.")
        globals()[name] = val
    define_ID_dependent_configs()


def import_configs(filename, overwritten_configs=None, create_dirs=False):
    overwritten_configs = dict(overwritten_configs or {})
    for name in overwritten_configs:
        if name not in configs_name:
            raise ValueError(f"The overwritten config '{name}' is not defined.")
    base, ext = os.path.splitext(filename)
    if ext == '':
        filename += '.py'
    elif ext != '.py':
        raise ValueError('Filename must be a .py file')
    path = filename if os.path.isabs(filename) else os.path.join(globals()['configs_dir'], filename)
    assert os.path.exists(path), f"The configuration file {filename} does not exist in {globals()['configs_dir']}"
    spec = importlib.util.spec_from_file_location('user.config', path)
    user = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(user)
    values = {}
    for name, val in vars(user).items():
        if isinstance(val, (ModuleType, FunctionType)) or name.startswith('__'):
            continue
        if name not in configs_name:
            raise ValueError(f"The imported config '{name}' is not defined.")
        values[name] = val
    values.update(overwritten_configs)
    globals().update(values)
    define_ID_dependent_configs()
    validate_configs(create_dirs=create_dirs)


define_ID_dependent_configs()

if __name__ == '__main__':
    print_configs()
