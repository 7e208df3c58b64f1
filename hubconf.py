"""torch.hub entry point with the reference's name and arguments (reference hubconf.py:17-48):
``torch.hub.load(<repo>, 'mars5_english', source='local')`` -> (Mars5TTS, InferenceConfig)."""
dependencies = ['torch', 'numpy', 'safetensors', 'regex']

import logging
import os

import torch

from inference import InferenceConfig, Mars5TTS

ar_url = "https://github.com/Camb-ai/MARS5-TTS/releases/download/v0.4/mars5_en_checkpoints_ar-3000000.pt"
nar_url = "https://github.com/Camb-ai/MARS5-TTS/releases/download/v0.3/mars5_en_checkpoints_nar-1980000.pt"
ar_sf_url = "https://github.com/Camb-ai/MARS5-TTS/releases/download/v0.4/mars5_en_checkpoints_ar-3000000.safetensors"
nar_sf_url = "https://github.com/Camb-ai/MARS5-TTS/releases/download/v0.3/mars5_en_checkpoints_nar-1980000.safetensors"


def _fetch(url: str, fmt: str, progress: bool) -> dict:
    if fmt == 'pt':
        return torch.hub.load_state_dict_from_url(url, progress=progress, check_hash=False, map_location='cpu')
    from safetensors import safe_open
    ckpt_dir = os.path.join(torch.hub.get_dir(), 'checkpoints')
    os.makedirs(ckpt_dir, exist_ok=True)
    cached = os.path.join(ckpt_dir, os.path.basename(torch.hub.urlparse(url).path))
    if not os.path.exists(cached):
        torch.hub.download_url_to_file(url, cached, None, progress=progress)
    ckpt = {'model': {}}
    with safe_open(cached, framework='pt', device='cpu') as f:
        md = f.metadata()
        ckpt['vocab'] = {'texttok.model': md['texttok.model'], 'speechtok.model': md['speechtok.model']}
        for k in f.keys():
            ckpt['model'][k] = f.get_tensor(k)
    return ckpt


def mars5_english(pretrained=True, progress=True, device=None, ckpt_format='safetensors', ar_path=None, nar_path=None):
    """ Load mars5 english model on `device`, optionally show `progress`. """
    if device is None:
        device = 'cuda' if torch.cuda.is_available() else 'cpu'
    assert ckpt_format in ['safetensors', 'pt'], "checkpoint format must be 'safetensors' or 'pt'"
    logging.info(f"Using device: {device}")
    if pretrained == False:   # noqa: E712  (same check as the reference)
        raise AssertionError('Only pretrained model currently supported.')
    ar_ckpt = torch.load(str(ar_path), map_location='cpu') if ar_path is not None else \
        _fetch(ar_sf_url if ckpt_format == 'safetensors' else ar_url, ckpt_format, progress)
    nar_ckpt = torch.load(str(nar_path), map_location='cpu') if nar_path is not None else \
        _fetch(nar_sf_url if ckpt_format == 'safetensors' else nar_url, ckpt_format, progress)
    logging.info("Initializing modules...")
    return Mars5TTS(ar_ckpt, nar_ckpt, device=device), InferenceConfig
