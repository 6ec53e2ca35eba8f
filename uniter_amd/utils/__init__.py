"""Drop-in counterparts of the reference's ``utils`` helpers that sit on the training hot path
(utils/distributed.py, utils/misc.py) plus the flat parameter / gradient arena used on MI355X."""
