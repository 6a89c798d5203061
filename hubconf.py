"""torch.hub entry points with the names, signatures and defaults of the reference's hubconf.py (:11-31, :34-88).
``torch.hub.load('<this repo>', 'ball_detection', model_name='wasb', source='local')``.

The reference's default detector 'segformerpp_b2' lives in an un-vendored hub repo (KieDani/SegformerPlusPlus) and is
not built here: the default call raises the documented NotImplementedError naming the in-tree alternative
('wasb' / 'hrnet'); ``full_pipeline()`` uses the in-tree detectors on both sides of the agreement filter."""
dependencies = ['torch', 'numpy']

import os  # noqa: E402

from upliftingtabletennis_amd.interface import BallDetector, TableDetector, TableTennisPipeline, UpliftingModel  # noqa: E402,F401

IMAGES_ZIP_URL = "https://mediastore.rz.uni-augsburg.de/get/51XbRH38ZY/"       # reference hubconf.py:8
IMAGES_ZIP_FILENAME = "example_images.zip"


def ball_detection(model_name='segformerpp_b2', **kwargs):
    """Loads the ball detection model.  Built here: 'wasb' (the in-tree HRNet); 'segformerpp_*' raises NotImplementedError."""
    return BallDetector(model_name=model_name, **kwargs)


def table_detection(model_name='segformerpp_b2', **kwargs):
    """Loads the table detection model.  Built here: 'hrnet' (the in-tree HRNet); 'segformerpp_*' raises NotImplementedError."""
    return TableDetector(model_name=model_name, **kwargs)


def full_pipeline():
    """Loads the end-to-end pipeline (ball + table detection, refine, uplift)."""
    return TableTennisPipeline()


def download_example_images(local_folder='example_images'):
    """Reference hubconf.py:34-88: return `local_folder` when it already holds the example images, otherwise fetch and
    unpack the archive.  Failures (no network on an air-gapped MI355X box) surface as the reference's RuntimeError."""
    import zipfile
    import torch
    if os.path.isdir(local_folder) and os.listdir(local_folder):
        return local_folder
    os.makedirs(local_folder, exist_ok=True)
    archive = os.path.join(local_folder, IMAGES_ZIP_FILENAME)
    if not os.path.exists(archive):
        try:
            torch.hub.download_url_to_file(IMAGES_ZIP_URL, archive, progress=True)
        except Exception as e:
            if os.path.exists(archive):
                os.remove(archive)
            raise RuntimeError(f"Failed to download images: {e}")
    try:
        with zipfile.ZipFile(archive, 'r') as z:
            z.extractall(local_folder)
    except Exception as e:
        raise RuntimeError(f"Failed to extract images: {e}")
    if os.path.exists(archive):
        os.remove(archive)
    return local_folder
