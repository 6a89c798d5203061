"""torch.hub entry points with the names and signatures of the reference's hubconf.py (:11-31).
``torch.hub.load('<this repo>', 'ball_detection', model_name='wasb', source='local')``."""
dependencies = ['torch', 'numpy']

from upliftingtabletennis_amd.interface import BallDetector, TableDetector, TableTennisPipeline, UpliftingModel  # noqa: E402,F401


def ball_detection(model_name='wasb', **kwargs):
    """Loads the ball detection model.  Built here: 'wasb' (the in-tree HRNet); 'segformerpp_*' raises."""
    return BallDetector(model_name=model_name, **kwargs)


def table_detection(model_name='hrnet', **kwargs):
    """Loads the table detection model.  Built here: 'hrnet' (the in-tree HRNet); 'segformerpp_*' raises."""
    return TableDetector(model_name=model_name, **kwargs)


def full_pipeline():
    """Loads the end-to-end pipeline (ball detection + refine + uplift)."""
    return TableTennisPipeline()
