"""Interface stub (see tests/stubs/README.md): the reading half of stable_baselines3.common.save_util as SB3 1.0 publishes it --
`json_to_data` (entries carrying ":serialized:" are base64 + cloudpickle) and `load_from_zip_file(path) -> (data, params,
pytorch_variables)` (every `*.pth` member except pytorch_variables.pth is a parameter dict named after the file).  Restated from the
published source for the round-trip test of drloco_amd.checkpoint.write_model_zip; NOT the package."""
import base64
import io
import json
import zipfile

import cloudpickle
import torch as th


def json_to_data(json_string, custom_objects=None):
    json_dict = json.loads(json_string)
    out = {}
    for key, item in json_dict.items():
        if custom_objects is not None and key in custom_objects:
            out[key] = custom_objects[key]
        elif isinstance(item, dict) and ':serialized:' in item:
            out[key] = cloudpickle.loads(base64.b64decode(item[':serialized:'].encode()))
        else:
            out[key] = item
    return out


def load_from_zip_file(load_path, load_data=True, custom_objects=None, device='cpu'):
    with zipfile.ZipFile(load_path) as archive:
        namelist = archive.namelist()
        data, pytorch_variables, params = None, None, {}
        if 'data' in namelist and load_data:
            data = json_to_data(archive.read('data').decode(), custom_objects=custom_objects)
        for file_path in [f for f in namelist if f.endswith('.pth')]:
            th_object = th.load(io.BytesIO(archive.read(file_path)), map_location=device, weights_only=False)
            if file_path in ('pytorch_variables.pth', 'tensors.pth'):
                pytorch_variables = th_object
            else:
                params[file_path[:-len('.pth')]] = th_object
    return data, params, pytorch_variables
