"""Small host-side helpers of the generation path."""
from datetime import datetime


def encode_time(timestamp):
    """ISO-8601 'YYYY-mm-ddTHH:MM:SSZ' -> minute-of-day code, weekends offset by 1440
    (reference: utils/data_util.py:30-39)."""
    t = datetime.strptime(timestamp, '%Y-%m-%dT%H:%M:%SZ')
    code = t.hour * 60 + t.minute
    return code + 1440 if t.weekday() >= 5 else code
