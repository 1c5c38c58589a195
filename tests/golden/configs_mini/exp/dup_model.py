model = dict(type="other")
